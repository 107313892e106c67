cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1m
(for P in 5; do MV_PROBE_LIB=tools/probe/libfbank_probe$P.so timeout 300 python tools/bench_fbank.py 2>&1 | tail -1; done) | tee gpurun_out/r1m/fbank_probe5.log
