#!/bin/bash
# quick GPU check: parity tests + benches, no profiling.  usage: bash tools/gpu_quick.sh <tag>
TAG=${1:-q}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; cd $REPO
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
for W in 8 12 16; do MV_FBANK_WAVES=$W timeout 300 python tools/bench_fbank.py >> $OUT/fbank_waves.log 2>&1; done; cat $OUT/fbank_waves.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-400
