cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1p
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4 | tee gpurun_out/r1p/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r1p/bench.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r1p/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r1p/rocprof.log 2>&1
for f in $(find $GRAFT_REPO_ROOT/gpurun_out/r1p/prof -name "*kernel_stats*.csv"); do head -12 $f | cut -c1-150; done
