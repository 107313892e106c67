cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1h
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r1h/pytest_gpu.log 2>&1; tail -2 gpurun_out/r1h/pytest_gpu.log
timeout 300 python tools/bench_fbank.py > gpurun_out/r1h/fbank.log 2>&1; tail -1 gpurun_out/r1h/fbank.log
timeout 900 python tools/bench_conv.py > gpurun_out/r1h/conv.log 2>&1; grep shape gpurun_out/r1h/conv.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r1h/bench.log 2>&1; tail -1 gpurun_out/r1h/bench.log | cut -c1-300
