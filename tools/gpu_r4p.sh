#!/bin/bash
# Round-4 session P: PMC passes of the split conv2d kernel on single layers (waves parked / issue-stalled / MFMA busy, LDS, traffic)
TAG=${1:-r12p}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
for shape in "s1 conv1" "s1 3x3" "s1 conv3" "s3 3x3" "s4 conv3"; do
  key=$(echo $shape | tr ' ' '_')
  for pass in sq1 sq2 mem; do
    case $pass in
      sq1) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16";;
      sq2) C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_SALU";;
      mem) C="FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE";;
    esac
    MV_BENCH_SHAPES="$shape" timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$key/$pass -o pmc -- python $REPO/tools/bench_conv2d.py 16 > $OUT/${key}_$pass.log 2>&1
  done
  echo "== $shape"; python $REPO/tools/pmc_summary.py $OUT/$key 2>&1 | grep -A40 "conv2ds_kernel" | head -34
done
