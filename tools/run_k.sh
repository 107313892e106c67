cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1k
for PAD in 0 64 96; do echo "padx $PAD"; MV_PADX=$PAD timeout 600 python tools/bench_conv.py 2>&1 | grep -E "c2c 1024|mfa 3072|asp 3072" ; done | tee gpurun_out/r1k/padx.log
for PAD in 0 64; do echo "probe2 (loads only) padx $PAD"; MV_PADX=$PAD MV_PROBE_LIB=tools/probe/libprobe2.so timeout 600 python tools/bench_conv.py 2>&1 | grep -E "c2c 1024|mfa 3072" ; done | tee -a gpurun_out/r1k/padx.log
