cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1z
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "asp_pool or ecapa or model" 2>&1 | tail -2
timeout 300 python tools/bench_asp.py 2>&1 | tail -2 | tee gpurun_out/r1z/asp.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity']['max_one_minus_cos'], d['roofline']['achieved'])"
