cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1s
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv1d or model or ecapa or end_to_end" 2>&1 | tail -3 | tee gpurun_out/r1s/tests.log
echo "asm-pipelined K stage (default build)" | tee gpurun_out/r1s/conv.log
timeout 600 python tools/bench_conv.py 2>&1 | grep -E "c2c|mfa" | grep '"tile": 256' | tee -a gpurun_out/r1s/conv.log
echo "previous kernel (old probe lib)" | tee -a gpurun_out/r1s/conv.log
MV_PROBE_LIB=tools/probe/libconv1d_probe0.so timeout 600 python tools/bench_conv.py 2>&1 | grep -E "c2c|mfa" | grep '"tile": 256' | tee -a gpurun_out/r1s/conv.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r1s/bench.log
