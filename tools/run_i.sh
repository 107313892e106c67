cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1i
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "conv1d" > gpurun_out/r1i/pytest_gpu.log 2>&1; tail -2 gpurun_out/r1i/pytest_gpu.log
timeout 900 python tools/bench_conv.py > gpurun_out/r1i/conv.log 2>&1; grep shape gpurun_out/r1i/conv.log
