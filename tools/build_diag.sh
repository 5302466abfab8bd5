#!/bin/bash
# development: build a GEMV_DIAG variant of the HIP module (in-kernel stamps) for ONE gemv translation unit.
# usage: tools/build_diag.sh gemv_q4k   -> llamafile_amd/libllamafile_amd_hip_diag.so  (use with LFAMD_HIP_SO=...)
set -e
TU=${1:-gemv_q4k}
cd "$(dirname "$0")/../llamafile_amd/csrc"
make -s -j8
mkdir -p diag
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=13 -DGEMV_DIAG=1 -c $TU.hip -o diag/$TU.o
OBJS=$(ls *.o | grep -v "^$TU.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libllamafile_amd_hip_diag.so diag/$TU.o $OBJS -ldl
echo built ../libllamafile_amd_hip_diag.so
