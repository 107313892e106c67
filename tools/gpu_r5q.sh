#!/bin/bash
# Round-5 session Q: what the driver runs at the end of the round, on the final tree -- the GPU suite, smoke(), `python3 bench.py --gpus 1 --steps 20
# --warmup 5` with every side leg -- plus the rocprofv3 kernel summary of the same headline command
TAG=${1:-r14q}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -x -q -m gpu --timeout 400 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log | cut -c1-300
timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?"; grep "^{" $OUT/bench.log | cut -c1-600
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ecapa -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/rocprof_ecapa.log 2>&1
f=$(find $OUT/prof_ecapa -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/rocprofv3_kernel_stats_ecapa1024.csv; rm -rf $OUT/prof_ecapa
head -8 $OUT/rocprofv3_kernel_stats_ecapa1024.csv | cut -c1-150
