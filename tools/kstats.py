"""Prints a rocprofv3 kernel_stats.csv compactly: name, calls, avg/min/max ns, %."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0][:40]
    print(f"{n:42s} calls={r['Calls']:>5s} avg={float(r['AverageNs']):10.0f} min={r['MinNs']:>8s} max={r['MaxNs']:>8s} pct={r['Percentage']}")
