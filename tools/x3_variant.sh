#!/bin/bash
# Dev tool (build box): A/B builds of one kernel source with extra -D flags -> build/libpfhip_<name>.so (loaded through PFHIP_LIB)
#   tools/x3_variant.sh <name> "<flags>" [source.hip, default gemm_x3.hip]
set -e
cd "$(dirname "$0")/../asr-2pass_amd/csrc"
OBJ=../../build/obj
SRC=${3:-gemm_x3.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $2 -c $SRC -o /tmp/${SRC%.hip}_$1.o
objs=$(ls $OBJ/*.o $OBJ/host/*.o | grep -v "$SRC.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/${SRC%.hip}_$1.o -o ../../build/libpfhip_$1.so
echo built build/libpfhip_$1.so
