#!/usr/bin/env python
"""Distribution of step-to-step intervals (gather kernel start to next gather start) from a kernel trace."""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
g = np.array([int(r['Start_Timestamp']) for r in rows if 'vae_gather_kernel' in r['Kernel_Name']], dtype=np.int64)
d = np.diff(g) / 1e3
print('steps', len(g), 'median', np.median(d), 'p10', np.percentile(d, 10), 'p90', np.percentile(d, 90), 'max', d.max())
big = np.flatnonzero(d > 2 * np.median(d))
print('intervals > 2x median:', len(big), 'sum of them (ms)', d[big].sum() / 1e3, 'sum all (ms)', d.sum() / 1e3)
print('first 60 deltas:', np.round(d[:60]).astype(int).tolist())
