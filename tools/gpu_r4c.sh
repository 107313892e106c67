#!/bin/bash
# Round-4 session C: Fbank occupancy variants (8 / 12 / 16 waves, lean registers): parity + timing; batch-1 forwards with the 64 x 64 conv tiles.
TAG=${1:-r12c}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== fbank variants"
for rep in 1 2; do
for v in w8 w12 w16; do
  timeout 300 python tools/check_fbank_variant.py tools/probe/libfbankw_$v.so 2>&1 | grep "^{\|Error\|error" | tee -a $OUT/fbank_variants.log
done
timeout 300 python tools/bench_fbank.py 2>&1 | grep "^{" | cut -c70-200 | tee -a $OUT/fbank_variants.log
done
echo "== batch 1 / small batches"
for m in ecapa1024 campp; do for B in 1 8; do timeout 300 python tools/bench_latency.py $m $B 50 2>&1 | grep "GPU time" | tee -a $OUT/latency.log; done; done
echo "== conv / model tests"; timeout 900 python -m pytest tests -m gpu -q -x -k "conv1d or native_model or backbones or campp_stress or predictor or trainer" -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/pytest_subset.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_b1 -o b1 -- python $REPO/tools/bench_latency.py ecapa1024 1 50 > $OUT/b1.log 2>&1
find $OUT/prof_b1 -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "head -12 {} | cut -c1-150"
