#!/bin/bash
# store cache-policy bits of the persistent conv epilogue: 1 nt, 2 sc0, 3 sc1, 4 sc0 sc1, 5 sc0 nt, 6 sc1 nt, 7 sc0 sc1 nt
OUT=gpurun_out/ab_store; mkdir -p $OUT; rm -f $OUT/conv.log
for rep in 1 2; do
for v in product st1 st2 st3 st4 st5 st6 st7; do
  if [ $v = product ]; then unset MV_PROBE_LIB; else export MV_PROBE_LIB=tools/probe/libconv1d_$v.so; fi
  echo "== $v rep $rep" >> $OUT/conv.log
  timeout 300 python tools/bench_conv.py 2>&1 | grep -E '"tile": 256' | grep -E "c2c 1024|mfa 3072|c2c 512" >> $OUT/conv.log
done
done
cat $OUT/conv.log
