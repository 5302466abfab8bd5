"""development: time the decode launches that fuse sibling mat-muls (attn q/k/v two-type launch; ffn gate+up)."""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from llamafile_amd import sgemm, synth, _hip, ggml_types as T
sgemm.init(0)
k = 4096
def bench(spec, copies, label):
    sets = [[sgemm.upload_weights(t, synth.random_weights_torch(t, m, k, seed=s * 5 + i), m, k) for i, (t, m) in enumerate(spec)]
            for s in range(copies)]
    B = torch.randn(1, k, device="cuda").view(torch.uint8).view(1, k * 4)
    def go():
        for Ws in sets:
            sgemm.mul_mat_multi(Ws, B, T.F32, n=1)
    go(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        go()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{label}: {e0.elapsed_time(e1) * 1e3 / (20 * copies):.2f} us/launch", flush=True)
bench(((T.Q4_K, 4096), (T.Q4_K, 1024), (T.Q6_K, 1024)), 40, "attn q/k/v  Q4_K+Q4_K+Q6_K (dual)")
bench(((T.Q4_K, 4096), (T.Q4_K, 1024), (T.Q4_K, 1024)), 40, "attn q/k/v  Q4_K x3")
bench(((T.Q4_K, 14336), (T.Q4_K, 14336)), 10, "ffn gate+up Q4_K x2")
