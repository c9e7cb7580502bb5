#!/bin/bash
# Dev tool (build box): timing-only ablation builds of gemm_x3.hip (see PFHIP_X3_ABLATE there) -> build/libpfhip_x3_ab<n>.so
set -e
cd "$(dirname "$0")/../asr-2pass_amd/csrc"
OBJ=../../build/obj
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DPFHIP_X3_ABLATE=$n -c gemm_x3.hip -o /tmp/gemm_x3_ab$n.o
  objs=$(ls $OBJ/*.o $OBJ/host/*.o | grep -v "gemm_x3.hip.o")
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/gemm_x3_ab$n.o -o ../../build/libpfhip_x3_ab$n.so
  echo built build/libpfhip_x3_ab$n.so
done
