cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1t
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv1d or model or ecapa or end_to_end" 2>&1 | tail -2
(echo "DMA issue interleaved with the MFMA steps"; timeout 600 python tools/bench_conv.py 2>&1 | grep -E "c2c|mfa" | grep '"tile": 256') | tee gpurun_out/r1t/conv_interleaved.log
timeout 300 python tools/trace_conv.py 1024 1024 2>&1 | tee gpurun_out/r1t/trace_c2c_interleaved.log | sed -n 18,40p
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity'], d['roofline']['achieved'])"; done
