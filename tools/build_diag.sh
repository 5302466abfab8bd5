#!/bin/bash
# development: build a diagnostic variant of the HIP module (in-kernel stamps) for ONE translation unit.
# usage: tools/build_diag.sh gemv_q4k                  (-DGEMV_DIAG=1, the decode GEMVs' stamps: tools/gemv_stamps.py)
#        tools/build_diag.sh gemm_lf -DLF_STAMPS=1     (the Q8_0 batch body's stamps: tools/lf_stamps.py)
#   -> llamafile_amd/libllamafile_amd_hip_diag.so  (use with LFAMD_HIP_SO=...)
set -e
TU=${1:-gemv_q4k}
shift || true
DEFS=${@:--DGEMV_DIAG=1}
cd "$(dirname "$0")/../llamafile_amd/csrc"
make -s -j8
mkdir -p diag
EXTRA=""
case $TU in
  gemv_*) EXTRA="-mllvm -amdgpu-kernarg-preload-count=13" ;;
  gemm_*) EXTRA="-fno-slp-vectorize" ;;
esac
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off $EXTRA $DEFS -c $TU.hip -o diag/$TU.o
OBJS=$(ls *.o | grep -v "^$TU.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libllamafile_amd_hip_diag.so diag/$TU.o $OBJS -ldl
echo built ../libllamafile_amd_hip_diag.so
