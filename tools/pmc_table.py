"""Pivots a rocprofv3 counter_collection.csv: one row per kernel (mean over its dispatches)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:28]
    agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted({c for k in agg.values() for c in k})
print('kernel'.ljust(30), 'n', *[c[-22:].rjust(23) for c in names])
for k, v in agg.items():
    n = len(next(iter(v.values())))
    sel = (lambda xs: min(xs)) if len(sys.argv) > 2 and sys.argv[2] == 'min' else (lambda xs: sum(xs) / len(xs))
    print(k.ljust(30), n, *[f"{sel(v[c]):23.0f}" if c in v else ' ' * 23 for c in names])
