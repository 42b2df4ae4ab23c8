#!/usr/bin/env python
"""Kernel timeline of one bf16 training step + per-kernel totals from a rocprofv3 kernel_trace.csv:
    python tools/gpu/gpu_timeline16.py kernel_trace.csv [k-th step]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
def short(n):
    s = n.split('(')[0].replace('void ', '').replace('vh::', '')
    if 'gemm_bf16_kernel' in n or 'gemm_f32_kernel' in n:
        s = 'gemm' + n[n.index('<'):n.index('>') + 1].replace(' ', '')
    return s
# a step ends with the optimiser's scalar tail; the row after it starts the next step.  (Until round 5 the batch gather marked
# the start of a step; with vae.prefetch_batch it runs on the side stream in the MIDDLE of the step before the one it feeds.)
ends = [i for i, n in enumerate(names) if 'vae_dadapt_finalize' in n]
if len(ends) < 8:   # (round 6: the tail rides on the last workgroup of the update kernel -- vae.fused_finalize -- which then ends a step)
    ends = [i for i, n in enumerate(names) if 'vae_dadapt16_kernel' in n]
idx = [i + 1 for i in ends[:-1]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
if any('vae_gather' in n for n in names[idx[k]:idx[k] + 2]) and k + 1 < len(idx) - 1:
    k += 1   # (prefer a step whose batch was prefetched: the first step of an epoch gathers on the main stream)
i0, i1 = idx[k], idx[k + 1]
t0 = int(rows[i0]['Start_Timestamp'])
prev_end = {}
for r in rows[i0:i1]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    q = r['Queue_Id']
    gap = (st - prev_end.get(q, t0)) / 1e3
    print(f"q{q:>2} start {(st - t0) / 1e3:8.2f}  dur {(en - st) / 1e3:7.2f}  gap {gap:6.2f}  {short(r['Kernel_Name'])[:90]}")
    prev_end[q] = en
print('step total us (first kernel start -> optimiser tail end)', (int(rows[i1 - 1]['End_Timestamp']) - t0) / 1e3,
      '; start -> next step start', (int(rows[i1]['Start_Timestamp']) - t0) / 1e3)
# totals over the steps [len/4, 3 len/4)
a, b = idx[len(idx) // 4], idx[3 * len(idx) // 4]
nsteps = 3 * len(idx) // 4 - len(idx) // 4
tot = collections.defaultdict(lambda: [0, 0.0])
for r in rows[a:b]:
    s = short(r['Kernel_Name'])
    tot[s][0] += 1
    tot[s][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print(f"--- per-step kernel time over {nsteps} steps")
allsum = 0.0
for s, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{s[:90]:90s} {c / nsteps:6.2f} calls/step  avg {t / c:7.2f} us  per-step {t / nsteps:7.2f} us")
    allsum += t / nsteps
print('sum of kernel time per step', allsum)
