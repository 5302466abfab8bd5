#!/bin/bash
# development helper: time the prefill GEMM with experimental builds of the HIP module
for v in base NO_XLOAD NO_WLOAD NO_COMPUTE; do
  if [ $v != base ]; then cp llamafile_amd/libllamafile_amd_hip.so /tmp/keep.so; cp llamafile_amd/exp_$v.so llamafile_amd/libllamafile_amd_hip.so; fi
  echo "== $v"; python tools/kbench.py --cases "Q4_K,4096,4096,512;Q4_K,14336,4096,512" --iters 5 2>&1 | grep Q4_K
  if [ $v != base ]; then cp /tmp/keep.so llamafile_amd/libllamafile_amd_hip.so; fi
done
