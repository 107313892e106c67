#!/bin/bash
# round 3, call j: PMC refresh (counters only + kernel trace, separate passes): FETCH / WRITE / TCC for the headline and for CAM++, SQ pass for CAM++;
# only the summaries travel back
TAG=${1:-r11pmc}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
RAW=/tmp/pmcraw
mkdir -p $OUT $RAW/e $RAW/c
cd /tmp && export TMPDIR=/tmp
run() { dir=$1; name=$2; args=$3; shift 3; timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $RAW/$dir/$name -o pmc -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs $args > $RAW/$dir/$name.log 2>&1; echo "$dir $name rc=$?"; }
run e fetch "" FETCH_SIZE
run e write "" WRITE_SIZE
run e tcc "" TCC_HIT_sum TCC_MISS_sum
run c fetch "--model campp" FETCH_SIZE
run c write "--model campp" WRITE_SIZE
run c tcc "--model campp" TCC_HIT_sum TCC_MISS_sum
run c sq1 "--model campp" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16
cd $REPO
python tools/pmc_summary.py $RAW/e > $OUT/pmc_summary.txt 2>&1
python tools/pmc_summary.py $RAW/c > $OUT/pmc_summary_campp.txt 2>&1
python tools/pmc_traffic.py $RAW/e $TAG > $OUT/pmc_traffic.log 2>&1
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
python tools/pmc_per_launch.py $RAW/e conv1d_ring_persistent_kernel 7 > $OUT/pmc_conv1d_ring_per_launch.log 2>&1
python tools/pmc_per_launch.py $RAW/c cam_dense_block_kernel 3 > $OUT/pmc_cam_dense_block_per_launch.log 2>&1
tail -4 $OUT/pmc_conv1d_ring_per_launch.log; tail -4 $OUT/pmc_cam_dense_block_per_launch.log
