#!/bin/bash
# Round-4 session AD: HBM traffic of single conv2ds launches (PMC passes over a three-launch command each)
TAG=${1:-r12ad}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
for layer in "s1 conv1" "s1 3x3" "s1 conv3" "s3 3x3"; do
  key=$(echo $layer | tr ' ' '_')
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$key/$c -o pmc -- python $REPO/tools/one_conv2ds_launch.py "$layer" > $OUT/${key}_$c.log 2>&1
  done
  grep "^layer" $OUT/${key}_FETCH_SIZE.log
  python - <<PY
import csv, glob
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob('$OUT/$key/' + c + '/*counter_collection.csv'):
        v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'conv2ds_kernel' in r['Kernel_Name'] and r['Counter_Name'] == c]
        if v:
            print('   %s: %d launches, mean %.4g (raw counter units), min %.4g max %.4g' % (c, len(v), sum(v) / len(v), min(v), max(v)))
PY
done
