#!/bin/bash
# Builds build/ab/libpfhip_old.so = the in-tree library with ONE source file taken from git HEAD instead of the working tree
# (the baseline of an A/B timing run, see tools/ab_bench.sh):   tools/ab_build.sh asr-2pass_amd/csrc/stream_fused.hip
# The in-tree library must be built first (its other objects are reused).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$1; BASE=$(basename "$SRC")
mkdir -p "$ROOT/build/ab"
TMP="$ROOT/asr-2pass_amd/csrc/_ab_${BASE%.*}.hip"
git -C "$ROOT" show "HEAD:$SRC" > "$TMP"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -x hip -c "$TMP" -o "$ROOT/build/ab/old_$BASE.o"
rm -f "$TMP"
OBJS=$(ls "$ROOT"/build/obj/*.o "$ROOT"/build/obj/host/*.o | grep -v "/$BASE.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS "$ROOT/build/ab/old_$BASE.o" -o "$ROOT/build/ab/libpfhip_old.so"
echo "built build/ab/libpfhip_old.so with HEAD:$SRC"
