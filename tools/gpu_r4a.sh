#!/bin/bash
# Round-4 session A: all GPU tests on the new Fbank forms (reentrancy, batch-size invariance, fp64 arbiter), Fbank micro-benchmark A/B
# (product vs the round-3 kernel, phase probes, in-kernel timeline), small-batch forms, one short headline line.
TAG=${1:-r12a}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
rocminfo | grep -E "Marketing Name|Compute Unit" | head -4 > $OUT/rocminfo.txt 2>&1; nproc >> $OUT/rocminfo.txt
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -q -rA -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|Error|FAILED|fbank 256" $OUT/pytest_gpu.log | tail -12
echo "== fbank A/B (256 x 3 s, one workgroup per utterance)"
for rep in 1 2; do
  timeout 300 python tools/bench_fbank.py 2>&1 | grep "^{" | tee -a $OUT/fbank_ab.log
  MV_PROBE_LIB=tools/probe/libfbank_r3.so timeout 300 python tools/bench_fbank.py 2>&1 | grep "^{" | tee -a $OUT/fbank_ab.log
done
for v in base noload nomel nofft notile occ4; do
  MV_PROBE_LIB=tools/probe/libfbankp_$v.so timeout 300 python tools/bench_fbank.py 2>&1 | grep "^{" | tee -a $OUT/fbank_probes.log
done
echo "== timeline"; timeout 300 python tools/probe_fbank_phases.py run 2>&1 | grep -v amdgpu.ids | tee $OUT/fbank_timeline.log
echo "== small batches (product call with workspace vs one workgroup per utterance)"
for B in 1 8 32 128; do
  MV_BENCH_WS=1 timeout 300 python tools/bench_fbank.py $B 2>&1 | grep "^{" | cut -c1-200 | tee -a $OUT/fbank_small.log
  timeout 300 python tools/bench_fbank.py $B 2>&1 | grep "^{" | cut -c1-200 | tee -a $OUT/fbank_small.log
done
echo "== bench (headline only)"; timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $OUT/bench.log 2>&1; echo "bench rc=$?"
grep "^{" $OUT/bench.log | cut -c1-1500
