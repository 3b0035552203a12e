# Kernel trace of the bench's sequence workload: per-kernel stats and the per-dispatch durations of the
# search kernels in launch order (iteration 0 of every step searches all rings).
#   tools/trace_seq.sh <tag> [workload]
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_$1
mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py --workload ${2:-sequence} --no-cpu-baseline --no-extras --no-latency --steps 3 --warmup 1 > $O/kt.log 2>&1
cd $R
python tools/rocpd_export.py stats $O/kt/kt_results.db $O/kernel_stats.csv
python tools/kstats.py $O/kernel_stats.csv > $O/kstats.txt
sed -n 1,12p $O/kstats.txt
python - <<PY
import sqlite3
con = sqlite3.connect("$O/kt/kt_results.db")
rows = con.execute("select name, start, duration from kernels order by start").fetchall()
for key in ("k_ls_batch", "k_ls_first", "k_ls_items", "k_ls_merge"):
    ls = [d for n, s, d in rows if key in n]
    if ls:
        print(key, len(ls), "launches; us in launch order, first 20:", [round(d / 1e3) for d in ls[:20]])
PY
grep '^{' $O/kt.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
