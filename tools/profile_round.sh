set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/p20
mkdir -p $O
cd $R
timeout -k 10 400 python bench.py > $O/bench_line.json 2> $O/bench.err
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 1 > $O/kt.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o f -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/f.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/write -o w -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/w.log 2>&1
cd $R
timeout -k 10 100 python tools/bench_hbsum.py > $O/hbsum.txt 2>&1
cat $O/bench_line.json
