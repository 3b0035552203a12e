# Kernel trace of the device chain over the thirty real sweeps (tools/bench_chain.py): written under gpurun_out/prof_chain,
# the summary copied into profiles/ by hand.
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_chain
mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/chain -o chain -- python3 $R/tools/bench_chain.py 6 > $O/chain.log 2>&1
cd $R
python tools/rocpd_export.py stats $O/chain/chain_results.db $O/chain_kernel_stats.csv
python tools/kstats.py $O/chain_kernel_stats.csv > $O/chain_kstats.txt; sed -n 1,40p $O/chain_kstats.txt; cat $O/chain.log | tail -8
