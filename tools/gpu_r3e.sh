#!/bin/bash
# round 3, call e: melspec_pow2_kernel waves per workgroup A/B (4 = prefetch form, 8, 12), the Fbank stage-2-on-MFMA timing probe
# (tools/probe_fbank_mfma.py), front-end + model-info tests.  usage: bash tools/gpu_r3e.sh <tag>
TAG=${1:-r09a}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
for w in 4 8 12; do
  MV_MELFFT_WAVES=$w timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "melspec or featurizer" > $OUT/pytest_mel_w$w.log 2>&1; echo "pytest melspec waves=$w rc=$?"; tail -1 $OUT/pytest_mel_w$w.log
done
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "model_info or fbank" > $OUT/pytest_misc.log 2>&1; echo "pytest misc rc=$?"; tail -1 $OUT/pytest_misc.log
for i in 1 2; do
  for w in 4 8 12; do
    for which in readme 512 256; do
      echo -n "waves=$w " | tee -a $OUT/melfft.log; MV_MELFFT_WAVES=$w timeout 300 python tools/bench_melspec.py 256 48000 $which 2>&1 | grep kernel | tee -a $OUT/melfft.log
    done
  done
done
timeout 300 python tools/bench_melspec.py 2>&1 | grep kernel | tee -a $OUT/melfft.log
for i in 1 2; do
  for v in base mfma2 nostage2; do
    MV_PROBE_LIB=$REPO/tools/probe/libfbank_$v.so timeout 300 python tools/bench_fbank.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/fbank_probe.log
  done
done
timeout 300 python tools/bench_fbank.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/fbank_probe.log
