#!/bin/bash
# round 3, call f: head.conv1 inside the first FCM block's kernel -- layer tests, micro A/B, CAM++ end to end A/B (MV_FCM_C1=0 | 1), kernel stats
TAG=${1:-r09b}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "fcm or campp" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 300 python tools/bench_fcm_c1.py 2>&1 | grep "us per launch" | tee $OUT/fcm_c1_micro.log
for c1 in 0 1 0 1; do MV_FCM_C1=$c1 timeout 300 python bench.py --model campp --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('MV_FCM_C1=$c1', j['value'], j['ms_per_step'], j['parity'])" | tee -a $OUT/ab.log; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --model campp --steps 5 --warmup 2 --no-cpu-baseline > $OUT/rocprof.log 2>&1
head -12 $OUT/prof/bench_kernel_stats.csv | cut -c1-200
