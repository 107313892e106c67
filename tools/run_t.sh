cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1v
(echo "stagger 1 (default build)"; timeout 600 python tools/bench_conv.py 2>&1 | grep -E "c2c|mfa" | grep '"tile": 256'
echo "no stagger"; MV_PROBE_LIB=tools/probe/libconv1d_probe0.so timeout 600 python tools/bench_conv.py 2>&1 | grep -E "c2c|mfa" | grep '"tile": 256'
echo "stagger 2"; MV_PROBE_LIB=tools/probe/libconv1d_probe4.so timeout 600 python tools/bench_conv.py 2>&1 | grep -E "c2c|mfa" | grep '"tile": 256') | tee gpurun_out/r1v/conv_stagger.log
