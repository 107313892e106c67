cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1y
(echo "8 steps (product)"; timeout 600 python tools/bench_conv.py 2>&1 | grep -E "c2c 1024|mfa 3072" | grep '"tile": 256'
for N in 4 2 1; do echo "transfers spread over $N steps"; MV_PROBE_LIB=tools/probe/libconv1d_probe1$N.so timeout 600 python tools/bench_conv.py 2>&1 | grep -E "c2c 1024|mfa 3072" | grep '"tile": 256'; done) | tee gpurun_out/r1y/conv_dma_steps.log
