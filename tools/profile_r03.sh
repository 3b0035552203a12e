# The profiles/r03_* summaries in one GPU call (written under gpurun_out/prof3, copied into profiles/ by hand):
# bench line; kernel trace stats of the headline workload, of the device chain, of the voxel sweep, of the cell grid and of
# the stand-alone NN calls; FETCH_SIZE / WRITE_SIZE passes (separate runs) with their calibration; SQ / TA / L2 passes.
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof3
mkdir -p $O
cd $R
timeout -k 10 500 python bench.py > $O/bench_line.json 2> $O/bench.err
B="python3 $R/bench.py --no-cpu-baseline --no-extras --no-latency --steps 3 --warmup 1"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-latency --steps 5 --warmup 1 > $O/kt.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/chain -o chain -- python3 $R/tools/bench_chain.py 6 > $O/chain.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/vox -o vox -- python3 $R/tools/bench_voxel_sweep.py 10 > $O/vox.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/grid -o grid -- python3 $R/tools/bench_grid.py > $O/grid.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/nn -o nn -- python3 $R/tools/latency_nn.py > $O/nn.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o f -- $B > $O/f.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/write -o w -- $B > $O/w.log 2>&1
timeout -s KILL 100 rocprofv3 --pmc FETCH_SIZE -d $O/cf -o f -- $R/tools/_build/calib_traffic > $O/cf.log 2>&1
timeout -s KILL 100 rocprofv3 --pmc WRITE_SIZE -d $O/cw -o w -- $R/tools/_build/calib_traffic > $O/cw.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d $O/sq -o p -- $B > $O/sq.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE -d $O/ta -o p -- $B > $O/ta.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/l2 -o p -- $B > $O/l2.log 2>&1
cd $R
python tools/rocpd_export.py stats $O/kt/kt_results.db $O/seq_kernel_stats.csv
for t in chain vox grid nn; do python tools/rocpd_export.py stats $O/$t/${t}_results.db $O/${t}_kernel_stats.csv; done
for t in fetch:f:pmc_FETCH_SIZE write:w:pmc_WRITE_SIZE cf:f:calib_FETCH_SIZE cw:w:calib_WRITE_SIZE sq:p:pmc_sq ta:p:pmc_ta l2:p:pmc_l2; do
  IFS=: read d o name <<< "$t"
  python tools/rocpd_export.py pmc $O/$d/${o}_results.db $O/$name.csv
done
python tools/make_traffic.py $O/bench_line.json $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv $O/calib_FETCH_SIZE.csv $O/calib_WRITE_SIZE.csv $O/traffic.json > /dev/null
P=$(python -c "import json; print(json.loads(open('$O/bench_line.json').read().strip().splitlines()[-1])['config']['pairs_per_gpu_per_step'])")
python tools/make_issue_pipes.py $O/pmc_sq.csv $O/pmc_ta.csv $O/pmc_l2.csv $O/issue_pipes.json r03 $P
python tools/kstats.py $O/seq_kernel_stats.csv > $O/kstats.txt; sed -n 1,8p $O/kstats.txt
cp $O/chain.log $O/chain.txt; cp $O/vox.log $O/voxel_sweep.txt; cp $O/grid.log $O/grid.txt; cp $O/nn.log $O/nn_calls.txt
rm -rf $O/kt $O/chain $O/vox $O/grid $O/nn $O/fetch $O/write $O/cf $O/cw $O/sq $O/ta $O/l2
ls $O
