cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1z
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for m in campp ecapa512; do timeout 600 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:20], d['value'], d['ms_per_step'], d['parity']['max_one_minus_cos'], d['roofline']['achieved'])"; done
