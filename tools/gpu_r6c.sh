#!/bin/bash
# Round-6 session C: (1) full GPU suite on the tree with MvConv1dDesc.clock_probe; (2) Fbank device A/Bs: mel power-row pitch
# FBT_PSTR 292 (product) / 272 / 304 (the two pitches the bank simulation of the mel stage's ds_read_b128 rates best), odd workgroups started ~4 / ~8 us
# late (does the 24 MB store burst of the tail stagger?), alternating, then SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE / SQ_WAIT_ANY of product and p272;
# (3) the ASP head's two passes over x on batch slices (Infinity Cache); (4) bench line with the box block's in-kernel ring clock.
TAG=${1:-r15c}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -q -m gpu --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log | cut -c1-300
for rep in 1 2 3; do
  for lib in product fb_p272 fb_p304 fb_late1 fb_late2; do
    if [ $lib = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=$REPO/tools/probe/lib$lib.so; fi
    timeout 120 python tools/bench_fbank.py 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib', $rep, d['us'], d['frac_of_8TBps'])" | tee -a $OUT/bench_fbank_ab.log
  done
done
unset MV_PROBE_LIB
timeout 300 python tools/bench_asp_chain.py 2>&1 | grep "^{" | tee $OUT/bench_asp_chain.log
timeout 600 python bench.py --no-cpu-baseline --no-other-configs > $OUT/bench_quick.log 2> $OUT/bench_quick.err; python - <<PY
import json
d = json.loads([l for l in open('$OUT/bench_quick.log') if l.startswith('{')][-1])
print('headline', d['value'], d['ms_per_step'], 'conv frac', d['roofline']['frac'], 'fbank', d['roofline_fbank']['avg_launch_us'])
print('box', {k: v for k, v in d['box'].items() if k != 'note'})
PY
cd /tmp && export TMPDIR=/tmp
for lib in product fb_p272; do
  if [ $lib = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=$REPO/tools/probe/lib$lib.so; fi
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_${lib}_a -o pmc -- python $REPO/tools/bench_fbank.py > $OUT/pmc_${lib}_a.log 2>&1; echo "pmc a $lib rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_${lib}_b -o pmc -- python $REPO/tools/bench_fbank.py > $OUT/pmc_${lib}_b.log 2>&1; echo "pmc b $lib rc=$?"
done
unset MV_PROBE_LIB
cd $REPO
python tools/pmc_summary.py $OUT > $OUT/pmc_summary_fbank.txt 2>&1; grep -A14 "fbank_tile" $OUT/pmc_summary_fbank.txt | head -70
find $OUT -name "*.csv" -size +200k -delete; rm -rf $OUT/pmc_*_a $OUT/pmc_*_b 2>/dev/null
