#!/usr/bin/env python3
"""Per-shape kernel timing on the GPU box (development tool).

Times lfamd_mul_mat for one (type, m, k, n) over `copies` distinct weight tensors visited round-robin,
so small matrices are streamed from HBM rather than from the 256 MiB Infinity Cache (SURVEY.md §7
"honest HBM roofline").  Reports device microseconds per launch (HIP events on the stream) and the
algorithmic GB/s or TFLOP/s.
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamafile_amd import _hip, ggml_types as T, sgemm, synth  # noqa: E402


EXTRA_FLAGS = 0


def run(tname, m, k, n, copies, iters, f32in=True, graph=True):
    t = T.BY_NAME[tname]
    L = _hip.lib()
    per = m * T.row_size(t, k)
    if copies <= 0:
        copies = max(1, min(64, int(600e6 // per) + 1))
    Ws = []
    raw = synth.random_weights_torch(t, m, k, 1)
    for c in range(copies):
        Ws.append(sgemm.upload_weights(t, raw, m, k))
    x = torch.rand((n, k), device="cuda") * 2 - 1
    vdt = T.VEC_DOT[t]
    if f32in:
        B, bt, brb = x, T.F32, k * 4
    else:
        B = sgemm.quantize_rows(vdt, x)
        bt, brb = vdt, B.stride(0)
    out = torch.empty((n, m), dtype=torch.float32, device="cuda")
    ws = torch.empty(max(16, sgemm.workspace_bytes(t, m, k, n)), dtype=torch.uint8, device="cuda")
    flags = sgemm.host_variant_flags() | EXTRA_FLAGS

    def launch_all():
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for W in Ws:
            rc = L.lfamd_mul_mat(t, C.c_void_p(W.data.data_ptr()), m, k, bt, C.c_void_p(B.data_ptr()), brb, n,
                                 C.c_void_p(out.data_ptr()), m, C.c_void_p(ws.data_ptr()), ws.numel(), flags, st)
            assert rc == 0, L.lfamd_last_error()

    launch_all()
    torch.cuda.synchronize()
    if graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            launch_all()
        fn = g.replay
    else:
        fn = launch_all
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (iters * copies)
    byts = per + n * k * 4 + n * m * 4
    flops = 2.0 * m * k * n
    print(f"{tname:6s} m={m:6d} k={k:6d} n={n:4d} copies={copies:3d}: {us:9.2f} us/launch  "
          f"{byts / us / 1e3:8.1f} GB/s  {flops / us / 1e6:8.1f} TFLOP/s", flush=True)
    return us


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--cases", default="decode")
    p.add_argument("--iters", type=int, default=20)
    p.add_argument("--copies", type=int, default=0)
    p.add_argument("--flags", type=int, default=0, help="extra LFAMD_FLAG_* bits (8 narrow, 16 wide, 32 plain, 2 precise)")
    p.add_argument("--prequant", action="store_true", help="activations already in vec_dot format")
    a = p.parse_args()
    EXTRA_FLAGS = a.flags
    sgemm.init(0)
    if a.cases == "decode":
        cases = [("Q4_K", 4096, 4096, 1), ("Q4_K", 1024, 4096, 1), ("Q4_K", 14336, 4096, 1), ("Q4_K", 4096, 14336, 1),
                 ("Q6_K", 1024, 4096, 1), ("Q6_K", 4096, 14336, 1), ("Q6_K", 128256, 4096, 1), ("Q8_0", 4096, 4096, 1),
                 ("Q8_0", 14336, 4096, 1), ("Q4_K", 4096, 4096, 4), ("Q4_K", 4096, 4096, 8)]
    elif a.cases == "prefill":
        cases = [("Q4_K", 4096, 4096, 512), ("Q4_K", 1024, 4096, 512), ("Q4_K", 14336, 4096, 512),
                 ("Q4_K", 4096, 14336, 512), ("Q6_K", 4096, 14336, 512), ("Q6_K", 128256, 4096, 512)]
    else:
        cases = []
        for c in a.cases.split(";"):
            tn, m, k, n = c.split(",")
            cases.append((tn, int(m), int(k), int(n)))
    for tn, m, k, n in cases:
        run(tn, m, k, n, a.copies, a.iters, f32in=not a.prequant)
