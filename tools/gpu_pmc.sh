#!/bin/bash
# PMC passes (counters only, with --kernel-trace) over a short bench run.  usage: bash tools/gpu_pmc.sh <tag> [bench args]
TAG=${1:-pmc}; shift
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-box $BENCH_ARGS > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
BENCH_ARGS="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
run grbm GRBM_GUI_ACTIVE
ls $OUT/*/ | head -30
