set -e
cd $GRAFT_REPO_ROOT
for D in 2 4; do
  pids=""
  for d in $(seq 1 $D); do
    timeout -k 10 120 python -u tools/bench_chain.py 8 > gpurun_out/fleet_p${D}_$d.txt 2>&1 &
    pids="$pids $!"
  done
  for p in $pids; do wait $p; done
  echo "== $D processes"; for d in $(seq 1 $D); do tail -2 gpurun_out/fleet_p${D}_$d.txt | cut -c1-90; done
done
