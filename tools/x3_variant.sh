#!/bin/bash
# Dev tool (build box): A/B builds of gemm_x3.hip with extra -D flags -> build/libpfhip_<name>.so (loaded through PFHIP_LIB)
#   tools/x3_variant.sh <name> "<flags>"
set -e
cd "$(dirname "$0")/../asr-2pass_amd/csrc"
OBJ=../../build/obj
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $2 -c gemm_x3.hip -o /tmp/gemm_x3_$1.o
objs=$(ls $OBJ/*.o $OBJ/host/*.o | grep -v "gemm_x3.hip.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/gemm_x3_$1.o -o ../../build/libpfhip_$1.so
echo built build/libpfhip_$1.so
