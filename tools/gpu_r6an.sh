#!/bin/bash
# Round-6 session AN: MelSpectrogram's pad / pad_mode on the device (melspec_extend_kernel in front of the transform kernels): the front-end tests of the GPU suite
# and a device fuzz of the melspec family with the two options in the generator
TAG=${1:-r15an}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -q -m gpu --timeout 600 -k "melspec or fbank or featur or front" > $OUT/pytest_gpu_frontend.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu_frontend.log | cut -c1-300
timeout 1800 python tools/emu_fuzz.py --device gpu --seed 929 --jobs 4 melspec 400 > $OUT/fuzz_melspec.log 2>&1; echo "rc=$?"; grep -E "ok,|FAIL" $OUT/fuzz_melspec.log | cut -c1-400 | tail -12
