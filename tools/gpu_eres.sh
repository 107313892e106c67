#!/bin/bash
# ERes2Net family on the GPU: parity tests, bench lines, rocprof kernel stats
OUT=gpurun_out/eres; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "conv2d or tstp or eres2net" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
for m in eres2netv2 eres2net; do
  timeout 600 python bench.py --model $m --steps 5 --warmup 2 --cpu-sample 16 > $OUT/bench_$m.log 2>&1; tail -1 $OUT/bench_$m.log | cut -c1-1500
done
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o bench -- python $REPO/bench.py --model eres2netv2 --steps 3 --warmup 1 --no-cpu-baseline > $REPO/$OUT/rocprof.log 2>&1
cd $REPO; for f in $(find $OUT/prof -name "*kernel_stats*.csv"); do head -12 $f | cut -c1-150; done
