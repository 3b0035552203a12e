# Counter passes over the scan-to-map leg: tools/pmc_map.sh <tag> <counter> [...]
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc
TAG=$1
shift
mkdir -p $O
cd /tmp
timeout -s KILL 300 rocprofv3 --pmc "$@" -d $O/$TAG -o p -- python3 $R/tools/bench_map.py 100 quick > $O/$TAG.log 2>&1
cd $R
python tools/rocpd_export.py pmc $O/$TAG/p_results.db $O/$TAG.csv
grep -E "k_grid_nn32|k_map_tiles" $O/$TAG.csv
