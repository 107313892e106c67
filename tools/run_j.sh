cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1j
for P in 1 2; do echo "probe $P"; MV_PROBE_LIB=tools/probe/libprobe$P.so timeout 600 python tools/bench_conv.py 2>&1 | grep -E "c2c 1024|mfa 3072" ; done | tee gpurun_out/r1j/probe.log
