#!/bin/bash
# quick GPU check of the two front-end kernels: parity tests of the front-ends + micro-benchmarks (+ PMC of bench_fbank when asked)
TAG=${1:-q}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "fbank or melspec or featurizer or res2 or ecapa" > $OUT/pytest_fe.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_fe.log
for i in 1 2; do timeout 300 python tools/bench_fbank.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/fe.log; timeout 300 python tools/bench_melspec.py 2>&1 | grep impl | tee -a $OUT/fe.log; done
MV_FBANK_IMPL=generic timeout 300 python tools/bench_fbank.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/fe.log
timeout 300 python tools/bench_res2.py 2>&1 | grep "res2" | tee -a $OUT/fe.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $OUT/bench_quick.log 2>&1; tail -1 $OUT/bench_quick.log | cut -c1-1800
if [ -n "$2" ]; then
  cd /tmp && export TMPDIR=/tmp
  for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
    n=$(echo $pass | cut -d' ' -f1)
    timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$n -o pmc -- python $REPO/tools/bench_fbank.py > $OUT/pmc_$n.log 2>&1
  done
  cd $REPO; python tools/pmc_summary.py $OUT 2>/dev/null | grep -A12 "fbank_tile" | head -60
fi
