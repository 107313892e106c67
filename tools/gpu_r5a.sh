#!/bin/bash
# Round-5 session A: all GPU tests (restored front-end goldens + the 50-case kaldi.fbank argument sweep on both kernels), smoke, the default bench,
# bench.py --gpus 2 on the 1-GPU box (must end with the "ranks met, too few devices" message), Fbank kernel timings per instantiation,
# the fuzzer's generators against the product library on the device under a time limit
TAG=${1:-r14a}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
timeout 120 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.log 2>&1; echo "bench rc=$?"; grep "^{" $OUT/bench.log | tail -1 | cut -c1-1200
timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/bench_gpus2_on_one_device.log 2>&1; echo "bench --gpus 2 rc=$?"; grep -i "ranks met" $OUT/bench_gpus2_on_one_device.log
for cfg in "auto 25" "generic 25" "tile 20" "generic 20" "tile 30" "generic 30"; do
  set -- $cfg
  MV_BENCH_KERNEL=$1 MV_BENCH_FRAME_LENGTH=$2 timeout 120 python tools/bench_fbank.py 256 >> $OUT/fbank_kernels.log 2>&1
done
cat $OUT/fbank_kernels.log | cut -c1-300
MV_FUZZ_STREAM=1 timeout 420 python tools/emu_fuzz.py fbank 160 --device gpu --jobs 4 --seed 3 > $OUT/fuzz_gpu_fbank.log 2>&1; echo "fuzz fbank rc=$?"; grep "RESULT\|FAIL" $OUT/fuzz_gpu_fbank.log | head -20
MV_FUZZ_STREAM=1 timeout 600 python tools/emu_fuzz.py all 48 --device gpu --jobs 4 --seed 4 > $OUT/fuzz_gpu_all.log 2>&1; echo "fuzz all rc=$?"; grep "RESULT\|FAIL" $OUT/fuzz_gpu_all.log | head -40
