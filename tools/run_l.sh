cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1l
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv1d or model or ecapa" 2>&1 | tail -5 | tee gpurun_out/r1l/tests.log
echo "persistent (default)" | tee gpurun_out/r1l/conv.log
timeout 600 python tools/bench_conv.py 2>&1 | grep -E "c2c|mfa" | tee -a gpurun_out/r1l/conv.log
echo "non-persistent (MV_CONV_PERSIST_BLOCKS=0)" | tee -a gpurun_out/r1l/conv.log
MV_CONV_PERSIST_BLOCKS=0 timeout 600 python tools/bench_conv.py 2>&1 | grep -E "c2c|mfa" | tee -a gpurun_out/r1l/conv.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r1l/bench.log
MV_CONV_PERSIST_BLOCKS=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r1l/bench_nonpersist.log
