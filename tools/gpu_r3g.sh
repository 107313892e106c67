#!/bin/bash
# round 3, call g: walk-direction / batch-slice knobs of the EcapaTdnn forward (MV_WALK, MV_ASP_CHUNKS): end-to-end A/B in one process,
# then ONE kernel trace of the same schedule cut into per-variant per-kernel times
TAG=${1:-r10a}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
timeout 400 python tools/bench_walk.py 30 3 > $OUT/walk_ab.log 2>&1; echo "walk rc=$?"; cat $OUT/walk_ab.log | tail -25
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/walkprof -o walk -- python $REPO/tools/bench_walk.py 12 1 > $OUT/walk_rocprof.log 2>&1; echo "rocprof rc=$?"
python $REPO/tools/walk_trace_summary.py /tmp/walkprof 12 1 > $OUT/walk_kernels.log 2>&1; cat $OUT/walk_kernels.log | cut -c1-600
