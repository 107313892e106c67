cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1w
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r1w/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | tee gpurun_out/r1w/bench.log | cut -c1-200
