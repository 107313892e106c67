#!/bin/bash
# Round-6 session BU (third session): after the two fbank findings of r15bt (a band beyond Nyquist is now refused as torchaudio's get_mel_banks asserts; the kernel-vs-fp32-oracle
# check no longer uses the fp32 oracle where IT is > 1e-3 from the fp64 arbiter): the fbank family again with the seed that failed and two new ones, melspec, then the full GPU suite + smoke
TAG=${1:-r15bu}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
for seed in 2040 3050 3051; do
  timeout 1200 python tools/emu_fuzz.py --device gpu --seed $seed --jobs 4 fbank 600 > $OUT/fuzz_fbank_$seed.log 2>&1; echo "fuzz fbank seed $seed rc=$?"; grep -E "ok,|FAIL" $OUT/fuzz_fbank_$seed.log | cut -c1-400
done
timeout 1200 python tools/emu_fuzz.py --device gpu --seed 3052 --jobs 4 melspec,fcm_c1 600 > $OUT/fuzz_melspec_fcm_c1.log 2>&1; echo "fuzz melspec,fcm_c1 rc=$?"; grep -E "ok,|FAIL" $OUT/fuzz_melspec_fcm_c1.log | cut -c1-400
timeout 900 python -m pytest tests -q -m gpu --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log | cut -c1-200
