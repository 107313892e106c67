#!/bin/bash
# Round-5 PMC passes over the CAM++ leg (counters only, with --kernel-trace): LDS / VALU / MFMA activity of cam_dense_block_kernel after r14k-m
TAG=${1:-r14pmc}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- python $REPO/bench.py --model campp --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
run grbm GRBM_GUI_ACTIVE
cd $REPO; python tools/pmc_summary.py $OUT > $OUT/pmc_summary_campp.txt 2>&1; grep -A24 "cam_dense_block" $OUT/pmc_summary_campp.txt | head -30
find $OUT -name "*.csv" -size +200k -delete; rm -rf $OUT/sq1 $OUT/sq2 $OUT/grbm 2>/dev/null; ls $OUT | head
