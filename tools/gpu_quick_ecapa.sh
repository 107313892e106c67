#!/bin/bash
# Ecapa quick session: conv1d / asp / model parity tests, headline bench A/B of one environment switch.  usage: bash tools/gpu_quick_ecapa.sh <tag> <ENV_NAME> <value A> <value B>
TAG=${1:-r04m}; VAR=${2:-MV_ASP_FUSE_STATS}; VA=${3:-1}; VB=${4:-0}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "conv1d or asp or ecapa or tdnn or native_library" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for v in $VA $VB $VA $VB; do env $VAR=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs 2>&1 | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$VAR=$v', j['value'], j['ms_per_step'], j['stages_ms'] if 'stages_ms' in j else '', j['parity']['max_one_minus_cos'])" | tee -a $OUT/ab.log; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/rocprof.log 2>&1
head -16 $OUT/prof/bench_kernel_stats.csv | cut -c1-150
