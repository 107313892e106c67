#!/bin/bash
# Round-6 session AL: kaldi.fbank's use_energy / raw_energy / energy_floor / htk_compat on the device (fbank_energy_kernel behind the mel kernels): the Fbank tests of
# the GPU suite (argument sweep with the eight new cases) and a device fuzz of the fbank family with the energy options in the generator
TAG=${1:-r15al}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -q -m gpu --timeout 600 -k "fbank or featur or front" > $OUT/pytest_gpu_fbank.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu_fbank.log | cut -c1-300
timeout 1800 python tools/emu_fuzz.py --device gpu --seed 828 --jobs 4 fbank 400 > $OUT/fuzz_fbank.log 2>&1; echo "rc=$?"; grep -E "ok,|FAIL" $OUT/fuzz_fbank.log | cut -c1-400 | tail -12
