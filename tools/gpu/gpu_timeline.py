#!/usr/bin/env python
"""Print the kernel timeline of one training step from a rocprofv3 kernel_trace.csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# find the start of a step well into the run: the k-th gather kernel
idx = [i for i, n in enumerate(names) if 'vae_gather_kernel' in n]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
i0, i1 = idx[k], idx[k + 1]
t0 = int(rows[i0]['Start_Timestamp'])
prev_end = t0
for r in rows[i0:i1]:
    n = r['Kernel_Name']
    short = n.split('(')[0].replace('void ', '').replace('vh::', '')
    if 'gemm_f32' in n:
        short = 'gemm' + n[n.index('<'):n.index('>') + 1].replace(' ', '')
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"q{r['Queue_Id']:>2} start {(st - t0) / 1e3:8.2f}  dur {(en - st) / 1e3:7.2f}  gap {(st - prev_end) / 1e3:6.2f}  {short[:70]}")
    prev_end = max(prev_end, en)
print('step total us', (int(rows[i1]['Start_Timestamp']) - t0) / 1e3)
