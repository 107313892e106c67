#!/bin/bash
# A/B of the block-0 window conv inside one box (MV_BLOCK0_WINDOW=0 -> 5-tap form)
OUT=gpurun_out/ab_block0; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "window or ecapa or predictor" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for i in 1 2; do
  for w in 0 1; do
    MV_BLOCK0_WINDOW=$w timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_w${w}_$i.log 2>&1
    python - <<PY
import json
l=[x for x in open("$OUT/bench_w${w}_$i.log") if x.startswith("{")][-1]; j=json.loads(l)
print("window=$w run $i", j["value"], j["ms_per_step"], j.get("parity"))
PY
  done
done
