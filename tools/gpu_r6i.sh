#!/bin/bash
# Round-6 session I: PMC traffic of the headline step on the final tree (FETCH_SIZE / WRITE_SIZE in separate passes, bench.py --no-box) + SQ passes,
# and rocprofv3 --kernel-trace --stats of the driver's bench command
TAG=${1:-r15i}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash $REPO/tools/gpu_pmc.sh $TAG > $OUT/pmc_passes.log 2>&1; tail -8 $OUT/pmc_passes.log
cd $REPO
python tools/pmc_traffic.py $OUT $TAG | tail -2
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-box > $OUT/bench_under_rocprof.log 2>&1; echo "rocprof rc=$?"
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats_ecapa1024.csv; rm -rf $OUT/prof
find $OUT -name "*.csv" -size +300k -delete; for d in sq1 sq2 fetch write tcc grbm; do rm -rf $OUT/$d; done
ls $OUT
