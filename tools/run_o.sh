cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1o
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4 | tee gpurun_out/r1o/tests.log
timeout 600 python tools/bench_conv.py 2>&1 | grep -E "asp" | tee gpurun_out/r1o/conv.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r1o/bench.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r1o/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r1o/rocprof.log 2>&1
for f in $(find $GRAFT_REPO_ROOT/gpurun_out/r1o/prof -name "*kernel_stats*.csv"); do head -20 $f | cut -c1-150; done
