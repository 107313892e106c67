#!/bin/bash
# Round-4 session AA: the CAM++ exact head on the split-operand kernels -- parity (stress / borderline goldens, layer cases), its price against the fp16 head
TAG=${1:-r12aa}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "campp or conv2ds or fcm" > $OUT/pytest_campp.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_campp.log
timeout 400 python -c "
import json, sys, torch
sys.argv = ['bench.py']
import bench
dev = torch.device('cuda:0')
for head in (None, 'f32'):
    r = bench.short_run('campp', dev, 256, 5, 2, 256, head=head)
    print(json.dumps({k: r[k] for k in r if k in ('value', 'ms_per_step', 'fcm_head', 'parity', 'metric')}))
" > $OUT/bench_campp_heads.log 2>&1; echo "bench rc=$?"; grep "^{" $OUT/bench_campp_heads.log | cut -c1-600
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o campp -- python -c "
import sys, torch
sys.path[:0] = ['$REPO']
sys.argv = ['bench.py']
import bench
bench.short_run('campp', torch.device('cuda:0'), 256, 3, 1, 8, head='f32')
" > $OUT/rocprof.log 2>&1
for f in $(find $OUT/prof -name "*kernel_stats*.csv"); do head -8 $f | cut -c1-150; done
