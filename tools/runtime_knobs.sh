#!/bin/bash
# HIP runtime switches against the decode launch chain (tools/decode_mix_probe.py: three decode launches in turn, in one graph)
# usage (through gpurun): bash tools/runtime_knobs.sh > gpurun_out/runtime_knobs.txt
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout -k 10 120 python3 tools/decode_mix_probe.py --iters 40 2>&1 | grep -v amdgpu.ids; }
run A=0
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1
run ROC_USE_FGS_KERNARG=0
run DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
run DEBUG_HIP_GRAPH_BATCH_SIZE=256
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run A=1
