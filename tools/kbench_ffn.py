#!/usr/bin/env python3
"""development: the fused decode feed-forward launch (lfamd_ffn_block) against the separate launches it replaces, rotating
over `copies` weight sets so the matrices stream from HBM."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamafile_amd import _hip, ggml_types as T, sgemm, synth
sgemm.init(0)
k, n_ff, m = 4096, 14336, 4096
td = T.BY_NAME[sys.argv[1]] if len(sys.argv) > 1 else T.Q4_K
copies = 8
sets = []
for c in range(copies):
    sets.append((sgemm.upload_weights(T.Q4_K, synth.random_weights_torch(T.Q4_K, n_ff, k, 1), n_ff, k),
                 sgemm.upload_weights(T.Q4_K, synth.random_weights_torch(T.Q4_K, n_ff, k, 2), n_ff, k),
                 sgemm.upload_weights(td, synth.random_weights_torch(td, m, n_ff, 3), m, n_ff)))
x = torch.rand((1, k), device="cuda") * 2 - 1
x2 = torch.rand((1, n_ff), device="cuda") * 2 - 1
out = torch.empty((1, m), device="cuda")
ws = torch.empty(int(_hip.lib().lfamd_ffn_block_workspace(n_ff)), dtype=torch.uint8, device="cuda")

def fused():
    for Wg, Wu, Wd in sets:
        sgemm.ffn_block(Wg, Wu, Wd, x, out=out, ws=ws)

def separate():
    for Wg, Wu, Wd in sets:
        sgemm.mul_mat_multi([Wg, Wu], x.view(torch.uint8).view(1, k * 4), T.F32, n=1)
        sgemm.mul_mat(Wd, x2.view(torch.uint8).view(1, n_ff * 4), T.F32, n=1)

for name, fn in (("fused", fused), ("separate", separate)):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:9s} {T.NAMES[td]} down: {e0.elapsed_time(e1) * 1e3 / (20 * copies):8.2f} us per feed-forward block", flush=True)
print("check", _hip.lib().lfamd_ffn_block_check())
