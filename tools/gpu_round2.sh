#!/bin/bash
# Round-2 GPU session: DPP probe, parity tests, Fbank A/B (tile kernel vs generic kernel), bench (headline + other configs),
# rocprof kernel stats, PMC passes.  usage (repo root on the GPU box): bash tools/gpu_round2.sh <tag> [skip-pmc]
TAG=${1:-r03a}
REPO=$(cd $(dirname $0)/.. && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8 > $OUT/rocminfo.txt 2>&1; nproc >> $OUT/rocminfo.txt
[ -x tools/probe/dpp_probe ] && tools/probe/dpp_probe > $OUT/dpp_probe.log 2>&1; cat $OUT/dpp_probe.log
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -rA -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|Error|FAILED" $OUT/pytest_gpu.log | tail -15
echo "== fbank A/B"
for impl in tile generic; do MV_FBANK_IMPL=$impl timeout 300 python tools/bench_fbank.py >> $OUT/fbank_ab.log 2>&1; done
for impl in tile generic; do MV_FBANK_IMPL=$impl timeout 300 python tools/bench_fbank.py >> $OUT/fbank_ab.log 2>&1; done
MV_FBANK_IMPL=tile timeout 300 python tools/bench_fbank.py 256 160000 >> $OUT/fbank_ab.log 2>&1
MV_FBANK_IMPL=generic timeout 300 python tools/bench_fbank.py 256 160000 >> $OUT/fbank_ab.log 2>&1
cat $OUT/fbank_ab.log
echo "== packed-complex arm"
for i in 1 2; do MV_PROBE_LIB=$REPO/tools/probe/libmvector_pk.so timeout 300 python tools/bench_fbank.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/fbank_ab.log; MV_PROBE_LIB=$REPO/tools/probe/libmvector_pk.so timeout 300 python tools/bench_melspec.py 2>&1 | grep impl | tee -a $OUT/melspec_ab.log; done
echo "== melspec A/B"
for impl in fft dft fft dft; do MV_MELSPEC_IMPL=$impl timeout 300 python tools/bench_melspec.py >> $OUT/melspec_ab.log 2>&1; done
grep impl $OUT/melspec_ab.log
echo "== res2"; timeout 300 python tools/bench_res2.py > $OUT/res2.log 2>&1; grep res2 $OUT/res2.log
echo "== smoke"; timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
echo "== bench"; timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; echo "bench rc=$?" | tee -a $OUT/bench.log
tail -2 $OUT/bench.log | cut -c1-3000
echo "== bench campp / ecapa512 / mel"
for m in campp ecapa512 ecapa512_mel eres2netv2; do timeout 600 python bench.py --model $m --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_$m.log 2>&1; grep "^{" $OUT/bench_$m.log | cut -c1-400; done
echo "== rocprof"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?" | tee -a $OUT/rocprof.log
for f in $(find $OUT/prof -name "*kernel_stats*.csv"); do head -25 $f; done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_campp -o bench -- python $REPO/bench.py --model campp --steps 5 --warmup 2 --no-cpu-baseline > $OUT/rocprof_campp.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mel -o bench -- python $REPO/bench.py --model ecapa512_mel --steps 5 --warmup 2 --no-cpu-baseline > $OUT/rocprof_mel.log 2>&1
cd $REPO
if [ -z "$2" ]; then echo "== pmc"; bash tools/gpu_pmc.sh $TAG/pmc; fi
