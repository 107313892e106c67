#!/bin/bash
# Round-6 session BK: the tree as it is left after the re-entry session's kernel changes (Res2Net chain straight-line requests, GEMM epilogue row check per wave,
# ASP statistics rows through LDS): the driver's commands (full GPU suite, smoke, python bench.py), then the PMC passes + rocprofv3 kernel stats (gpu_r6i.sh)
TAG=${1:-r15bk}
REPO=$(cd $(dirname $0)/.. && pwd)
bash $REPO/tools/gpu_r6g.sh $TAG
bash $REPO/tools/gpu_r6i.sh ${TAG}pmc
