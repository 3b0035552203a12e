import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
for r in con.execute("select name, count(*), avg(duration), min(duration) from kernels where name like '%k_ls_batch%' group by name").fetchall():
    print("   ", r[0].split("(")[0][-40:], r[1], "avg %.0f us  min %.0f us" % (r[2] / 1e3, r[3] / 1e3))
