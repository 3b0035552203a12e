"""Exports the summaries kept under profiles/ from rocprofv3's sqlite output:
    python tools/rocpd_export.py stats  <results.db> <out.csv>   kernel stats (calls, total/avg/min/max ns, %)
    python tools/rocpd_export.py pmc    <results.db> <out.csv>   per kernel and counter: dispatches, mean, min, max
"""
import csv
import sqlite3
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0]


def main():
    kind, db, out = sys.argv[1:4]
    con = sqlite3.connect(db)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        if kind == "stats":
            rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                               "from kernels group by name order by sum(duration) desc").fetchall()
            total = float(sum(r[2] for r in rows)) or 1.0
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
            for r in rows:
                w.writerow([short(r[0]), r[1], r[2], "%.1f" % r[3], r[4], r[5], "%.2f" % (100.0 * r[2] / total)])
        else:
            rows = con.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), "
                               "avg(grid_size) from counters_collection group by kernel_name, counter_name "
                               "order by sum(value) desc").fetchall()
            w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Mean", "Min", "Max", "MeanGridSize"])
            for r in rows:
                w.writerow([short(r[0]), r[1], r[2], "%.1f" % r[3], "%.1f" % r[4], "%.1f" % r[5], "%.0f" % r[6]])


if __name__ == "__main__":
    main()
