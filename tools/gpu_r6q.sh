#!/bin/bash
TAG=${1:-r15r}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
for rep in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], {k: v for k, v in d['box'].items() if k != 'note'})" | tee -a $OUT/box.log
done
