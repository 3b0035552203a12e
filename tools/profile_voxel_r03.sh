# Kernel trace of the voxel sweep (tools/bench_voxel_sweep.py): written under gpurun_out/prof3, the summary copied into profiles/ by hand.
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof3
mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/vox -o vox -- python3 $R/tools/bench_voxel_sweep.py 10 > $O/vox.log 2>&1
cd $R
python tools/rocpd_export.py stats $O/vox/vox_results.db $O/voxel_kernel_stats.csv
python tools/kstats.py $O/voxel_kernel_stats.csv > $O/voxel_kstats.txt; sed -n 1,40p $O/voxel_kstats.txt
