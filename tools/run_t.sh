cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "melspec or mel" 2>&1 | tail -2
timeout 600 python bench.py --model ecapa512_mel --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'], d['parity'])"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2b/prof3 -o bench -- python $GRAFT_REPO_ROOT/bench.py --model ecapa512_mel --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
for f in $(find $GRAFT_REPO_ROOT/gpurun_out/r2b/prof3 -name "*kernel_stats*.csv"); do grep -E "stft|cmn_mask|linear" $f | cut -c1-150; done
