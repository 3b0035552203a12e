# One hardware-counter pass over the bench: tools/pmc_bench.sh <tag> <counter> [<counter> ...]
# (each pass is a run of its own; summary in gpurun_out/pmc/<tag>.csv)
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc
TAG=$1
shift
mkdir -p $O
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc "$@" -d $O/$TAG -o p -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/$TAG.log 2>&1
cd $R
python tools/rocpd_export.py pmc $O/$TAG/p_results.db $O/$TAG.csv
grep k_ls_batch $O/$TAG.csv
