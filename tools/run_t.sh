cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2a
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "conv1d or ecapa or end_to_end" 2>&1 | tail -2
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity']['max_one_minus_cos'], d['roofline']['achieved'])"; done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2a/prof2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2a/rocprof.log 2>&1
for f in $(find $GRAFT_REPO_ROOT/gpurun_out/r2a/prof2 -name "*kernel_stats*.csv"); do head -8 $f | cut -c1-130; done
