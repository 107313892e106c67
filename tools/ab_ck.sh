#!/bin/bash
OUT=gpurun_out/ab_ck; mkdir -p $OUT
for rep in 1 2; do for ck in 0 16 32; do for m in eres2netv2 eres2net; do
  MV_CONV2D_CK=$ck timeout 600 python bench.py --model $m --steps 5 --warmup 2 --no-cpu-baseline > $OUT/b_${m}_$ck_$rep.log 2>&1
  python - <<PY
import json
l=[x for x in open("$OUT/b_${m}_$ck_$rep.log") if x.startswith("{")][-1]; j=json.loads(l)
print("$m ck=$ck rep $rep", j["value"], j["ms_per_step"])
PY
done; done; done
