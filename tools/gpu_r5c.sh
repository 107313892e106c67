#!/bin/bash
# Round-5 session C: GPU suite (fbank_tile_kernel<10/12/15>, MelSpectrogram argument sweep, persist_blocks_hint), smoke, the default bench (repeated
# side legs), rocprofv3 kernel stats of the headline and of CAM++, the FETCH / WRITE PMC passes for roofline.traffic, Fbank kernel timings per
# instantiation, the Fbank device fuzz
TAG=${1:-r14c}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -q -m gpu --durations=10 --timeout 300 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -16 $OUT/pytest_gpu.log | cut -c1-200
timeout 120 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.log 2>&1; echo "bench rc=$?"; grep "^{" $OUT/bench.log | tail -1 | cut -c1-600
for cfg in "auto 25" "tile 20" "tile 24" "tile 30" "tile 32" "generic 20"; do
  set -- $cfg
  MV_BENCH_KERNEL=$1 MV_BENCH_FRAME_LENGTH=$2 timeout 120 python tools/bench_fbank.py 256 2>/dev/null | grep "^{" >> $OUT/fbank_kernels.log
done
cat $OUT/fbank_kernels.log | cut -c1-260
MV_FUZZ_STREAM=1 timeout 300 python tools/emu_fuzz.py fbank 200 --device gpu --jobs 4 --seed 5 > $OUT/fuzz_gpu_fbank.log 2>&1; echo "fuzz fbank rc=$?"; grep "RESULT\|FAIL" $OUT/fuzz_gpu_fbank.log | head -12 | cut -c1-400
MV_FUZZ_STREAM=1 timeout 200 python tools/emu_fuzz.py melspec 80 --device gpu --jobs 4 --seed 6 > $OUT/fuzz_gpu_melspec.log 2>&1; echo "fuzz melspec rc=$?"; grep "RESULT\|FAIL" $OUT/fuzz_gpu_melspec.log | head -12 | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ecapa -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/rocprof_ecapa.log 2>&1
for f in $(find $OUT/prof_ecapa -name "*kernel_stats*.csv"); do head -8 $f | cut -c1-150; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_campp -o bench -- python $REPO/bench.py --model campp --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/rocprof_campp.log 2>&1
for f in $(find $OUT/prof_campp -name "*kernel_stats*.csv"); do head -8 $f | cut -c1-150; done
for c in FETCH_SIZE WRITE_SIZE; do
  sub=$(echo $c | tr A-Z a-z | sed 's/_size//')
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc/$sub -o pmc -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/pmc_$sub.log 2>&1; echo "pmc $c rc=$?"
done
ls $OUT/pmc/*/ | head
