#!/bin/bash
# Timing-probe libraries (never shipped): one .hip file rebuilt with -DMV_PROBE=<n>, the other objects reused from the
# product build.  usage: bash tools/build_probe.sh <file.hip> <n> ...   ->  tools/probe/lib<file>_probe<n>.so
set -e
REPO=$(cd $(dirname $0)/.. && pwd)
PKG=$REPO/voiceprintrecognition-pytorch_amd
SRC=$1; shift
mkdir -p $REPO/tools/probe
for P in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DNDEBUG -I $PKG/csrc -DMV_PROBE=$P $EXTRA_DEFS -x hip -c $PKG/csrc/$SRC -o $REPO/tools/probe/${SRC}_p$P.o
  OBJS=$(ls $PKG/build/*.o | grep -v "/$SRC.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $REPO/tools/probe/lib${SRC%.hip}_probe$P.so $OBJS $REPO/tools/probe/${SRC}_p$P.o
  echo built $REPO/tools/probe/lib${SRC%.hip}_probe$P.so
done
