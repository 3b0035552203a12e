# the all-rings pre-pass (k_first_rings*) at 4 (as built), 6 and 8 waves per SIMD: libraries built beforehand as
# tools/_build/libpcp_fr<W>.so with -DPCP_FIRST_RINGS_WAVES=<W>
cd $GRAFT_REPO_ROOT
for L in pointcloudprocessing_amd/libpcp.so tools/_build/libpcp_fr6.so tools/_build/libpcp_fr8.so; do
  echo "== $L"
  PCP_LIB_PATH=$GRAFT_REPO_ROOT/$L timeout -k 10 100 python tools/bench_chain.py 5 2>&1 | tail -2 | cut -c1-120
  PCP_LIB_PATH=$GRAFT_REPO_ROOT/$L timeout -k 10 100 python tools/latency_pair.py 2>&1 | grep "launch loop"
done
