# ASP pooling: parity tests + micro-benchmark (nomax line), optional probe libraries
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "asp" 2>&1 | tail -2
for lib in "" $ASP_PROBES ""; do MV_PROBE_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 120 python tools/bench_asp.py 2>&1 | grep nomax | cut -c1-300; done
