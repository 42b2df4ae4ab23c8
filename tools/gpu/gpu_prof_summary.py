#!/usr/bin/env python
"""Summarise a rocprofv3 kernel_stats.csv per training step: python tools/gpu/gpu_prof_summary.py stats.csv nsteps"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = float(sys.argv[2])
tot = 0
for r in rows:
    n = r['Name']
    if 'vae_' in n or 'gemm_f32' in n or 'fillBuffer' in n or 'copyBuffer' in n:
        short = n.split('(')[0].replace('void ', '').replace('vh::', '')
        if 'gemm' in n:
            short = n[n.index('gemm_f32_kernel'):n.index('>') + 1]
        calls = int(r['Calls']); avg = float(r['AverageNs']) / 1e3
        per = calls / nsteps * avg
        if 'gemm' in n or 'vae_' in n:
            tot += per
        print(f"{short:70s} calls/step {calls/nsteps:6.2f} avg {avg:7.2f} us  per-step {per:7.2f} us")
print('total vae+gemm per step', tot)
