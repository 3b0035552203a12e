set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/p24
mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/tools/bench_grid.py 50000000 > $O/kt.log 2>&1
cd $R
python tools/rocpd_export.py stats $O/kt/kt_results.db $O/grid50m_kernel_stats.csv
cat $O/grid50m_kernel_stats.csv
