# bench.py's headline with each of the libraries named on the command line (variant builds under tools/_build/):
#   bash tools/try_libs.sh pointcloudprocessing_amd/libpcp.so tools/_build/libpcp_x.so ...
cd $GRAFT_REPO_ROOT
for L in "$@"; do
  PCP_LIB_PATH=$GRAFT_REPO_ROOT/$L timeout -k 10 150 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras --no-latency > gpurun_out/try.json 2> gpurun_out/try.err || { tail -5 gpurun_out/try.err; exit 1; }
  python - "$L" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/try.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]), "pairs/s; hinted launch", round(d["roofline"]["avg_launch_us"]), "us, first", round(d["roofline"]["first_launch"]["avg_launch_us"]), "us")
PY
done
