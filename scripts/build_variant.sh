#!/bin/bash
# Build an alternative librelearn_hip.so with extra compiler flags for ONE translation unit (A/B timing of kernel
# variants; select it at run time with RELEARN_LIB=<path>).   usage: scripts/build_variant.sh <name> <unit.hip> <flags...>
set -eu
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; UNIT="$2"; shift 2
cd "$ROOT/relearn_amd/csrc"
make -s -j8
mkdir -p "$ROOT/scripts/probe/abl"
BASE=$(basename "$UNIT" .hip)
EXTRA=""
case "$BASE" in kernels_mfma|kernels_critic|kernels_seq*) EXTRA="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $EXTRA "$@" -c "$BASE.hip" -o "_build/$BASE.$NAME.o"
OBJS=$(ls _build/*.o | grep -v "\.[a-zA-Z0-9_]*\.o$" | grep -v "_build/$BASE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS "_build/$BASE.$NAME.o" -o "$ROOT/scripts/probe/abl/librelearn_$NAME.so" -ldl -lpthread
echo "$ROOT/scripts/probe/abl/librelearn_$NAME.so"
