#!/bin/bash
# end-to-end A/B of the conv output store policy (MV_CONV_STORE_NT = 0 never / 1 short K / 2 always)
OUT=gpurun_out/ab_store_e2e; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "conv or ecapa" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for rep in 1 2; do
for m in 0 1 2; do
  for model in ecapa1024 ecapa512 campp; do
  MV_CONV_STORE_NT=$m timeout 600 python bench.py --model $model --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_${model}_m${m}_$rep.log 2>&1
  python - <<PY
import json
l=[x for x in open("$OUT/bench_${model}_m${m}_$rep.log") if x.startswith("{")][-1]; j=json.loads(l)
print("$model nt=$m rep $rep", j["value"], j["ms_per_step"], j["roofline"]["achieved"], j.get("parity",{}).get("max_one_minus_cos"))
PY
  done
done
done
