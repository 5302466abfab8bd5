#!/usr/bin/env python3
"""Does a decode launch cost more when its neighbours in the stream are OTHER kernels?  (development probe)

Three decode shapes of the 8B model (attn_output 4096 x 4096, gate + up 28672 x 4096, ffn_down 4096 x 14336, all Q4_K, f32 row in),
`copies` weight tensors each so every launch streams from HBM.  One graph per shape (the same kernel back to back) and one graph
that takes them in the model's order; per-launch device time by HIP events.  If the mixed graph costs the sum of the three, a
launch does not care who ran before it."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamafile_amd import _hip, ggml_types as T, sgemm, synth  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--copies", type=int, default=16)
    p.add_argument("--iters", type=int, default=30)
    p.add_argument("--shapes", default="Q4_K,4096,4096;Q4_K,28672,4096;Q4_K,4096,14336")
    a = p.parse_args()
    sgemm.init(0)
    L = _hip.lib()
    flags = sgemm.host_variant_flags()
    shapes = []
    for c in a.shapes.split(";"):
        tn, m, k = c.split(",")
        shapes.append((T.BY_NAME[tn], int(m), int(k)))
    sets = []
    for t, m, k in shapes:
        raw = synth.random_weights_torch(t, m, k, 1)
        Ws = [sgemm.upload_weights(t, raw, m, k) for _ in range(a.copies)]
        x = torch.rand((1, k), device="cuda") * 2 - 1
        out = torch.empty((1, m), dtype=torch.float32, device="cuda")
        ws = torch.empty(max(16, sgemm.workspace_bytes(t, m, k, 1)), dtype=torch.uint8, device="cuda")
        sets.append((t, m, k, Ws, x, out, ws))

    def launch(s, c):
        t, m, k, Ws, x, out, ws = s
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = L.lfamd_mul_mat(t, C.c_void_p(Ws[c].data.data_ptr()), m, k, T.F32, C.c_void_p(x.data_ptr()), k * 4, 1,
                             C.c_void_p(out.data_ptr()), m, C.c_void_p(ws.data_ptr()), ws.numel(), flags, st)
        assert rc == 0, L.lfamd_last_error()

    def timed(body, launches):
        body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (a.iters * launches)

    alone = []
    for s in sets:
        us = timed(lambda s=s: [launch(s, c) for c in range(a.copies)], a.copies)
        alone.append(us)
        print(f"{T.NAMES[s[0]]} {s[1]} x {s[2]} alone: {us:.2f} us per launch", flush=True)
    mixed = timed(lambda: [launch(s, c) for c in range(a.copies) for s in sets], a.copies)
    print(f"one of each, in turn: {mixed:.2f} us per round; sum of the three alone: {sum(alone):.2f}", flush=True)
    # the same round twice as long (the graph's fixed cost, if any, halves)
    mixed2 = timed(lambda: [launch(s, c % a.copies) for c in range(2 * a.copies) for s in sets], 2 * a.copies)
    print(f"one of each, in turn, graph twice as long: {mixed2:.2f} us per round", flush=True)


if __name__ == "__main__":
    main()
