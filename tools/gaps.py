"""Busy time and gaps between consecutive kernels of a rocprofv3 kernel trace (sqlite):
    python tools/gaps.py <results.db> [first_kernel_substring]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
key = sys.argv[2] if len(sys.argv) > 2 else None
names = [r[0].replace("(anonymous namespace)::", "").split("(")[0] for r in rows]
# the last contiguous run of launches after the final k_normals (one resident pair loop)
last = max(i for i, n in enumerate(names) if "k_normals" in n)
seq = list(zip(names, rows))[last:last + 24]
t0 = seq[0][1][1]
busy = gap = 0
prev_end = None
for n, (_, s, e) in seq:
    g = 0 if prev_end is None else s - prev_end
    print(f"{n:28s} start {(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f} us  gap {g / 1e3:6.1f} us")
    busy += e - s
    gap += max(g, 0)
    prev_end = e
print(f"busy {busy / 1e3:.1f} us, gaps {gap / 1e3:.1f} us, span {(seq[-1][1][2] - t0) / 1e3:.1f} us")
