#!/usr/bin/env python
"""Average PMC counters per kernel name from a rocprofv3 counter_collection.csv."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r['Kernel_Name']
    if 'gemm_f32' in n:
        n = 'gemm' + n[n.index('<'):n.index('>') + 1].replace(' ', '')
    else:
        n = n.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').replace('vh::', '')
    agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n in sorted(agg):
    if len(sys.argv) > 2 and sys.argv[2] not in n:
        continue
    print(n, {k: round(sum(v) / len(v)) for k, v in agg[n].items()}, 'n=%d' % len(next(iter(agg[n].values()))))
