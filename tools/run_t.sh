cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2a
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 600 python tools/bench_conv.py 2>&1 | grep -E "c2c 1024|mfa 3072" | grep '"tile": 256'
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity']['max_one_minus_cos'], d['roofline']['achieved'])"; done
