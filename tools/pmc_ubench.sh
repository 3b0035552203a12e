# Calibrates the SQ occupancy counters on the gather micro-benchmark (known: 32 resident waves per CU).
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc
mkdir -p $O
cd /tmp
timeout -s KILL 100 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $O/ub -o p -- $R/tools/_build/ubench_gather > $O/ub.log 2>&1
cd $R
python tools/rocpd_export.py pmc $O/ub/p_results.db $O/ub.csv
cat $O/ub.csv
