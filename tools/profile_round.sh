# Produces the round's profile summaries in one GPU call: bench line, kernel trace stats,
# FETCH_SIZE and WRITE_SIZE passes (separate runs), the tile-tree HBM leg, grid and map legs.
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
cd $R
timeout -k 10 400 python bench.py > $O/bench_line.json 2> $O/bench.err
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 1 > $O/kt.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o f -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/f.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/write -o w -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/w.log 2>&1
cd $R
python tools/rocpd_export.py stats $O/kt/kt_results.db $O/kernel_stats.csv
python tools/rocpd_export.py pmc $O/fetch/f_results.db $O/pmc_FETCH_SIZE.csv
python tools/rocpd_export.py pmc $O/write/w_results.db $O/pmc_WRITE_SIZE.csv
python tools/make_traffic.py $O/bench_line.json $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv $O/traffic.json
timeout -k 10 100 python tools/bench_hbsum.py > $O/hbsum.txt 2>&1
timeout -k 10 200 python tools/bench_grid.py > $O/grid_1m.txt 2>&1
timeout -k 10 300 python tools/bench_grid.py 50000000 > $O/grid_50m.txt 2>&1
timeout -k 10 300 python tools/bench_map.py 100 > $O/map_12m.txt 2>&1
timeout -k 10 100 python tools/bench_voxel.py > $O/voxel.txt 2>&1
cat $O/bench_line.json
