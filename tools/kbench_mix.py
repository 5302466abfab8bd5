"""development: do decode kernels cost more per launch when DIFFERENT kernels alternate (as in a decode pass) than when the
same kernel repeats (tools/kbench.py)?  Three Q4_K shapes, each over rotating weight copies; graph-replayed."""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from llamafile_amd import sgemm, synth, _hip, ggml_types as T
sgemm.init(0)
L = _hip.lib()
shapes = [(4096, 4096), (14336, 4096), (4096, 14336)]
COP = 12
Ws = {s: [sgemm.upload_weights(T.Q4_K, synth.random_weights_torch(T.Q4_K, s[0], s[1], seed=i), s[0], s[1]) for i in range(COP)] for s in shapes}
xs = {k: torch.randn(1, k, device="cuda").view(torch.uint8).view(1, k * 4) for k in (4096, 14336)}
outs = {m: torch.empty((1, m), dtype=torch.float32, device="cuda") for m in (4096, 14336)}
def launch(s, i):
    sgemm.mul_mat(Ws[s][i], xs[s[1]], T.F32, n=1, out=outs[s[0]])
def timed(fn, n_launch):
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
    return e0.elapsed_time(e1) * 1e3 / 20 / n_launch
tot = 0.0
for s in shapes:
    t = timed(lambda: [launch(s, i) for i in range(COP)], COP)
    tot += t
    print(f"{s}: {t:.2f} us/launch repeated", flush=True)
t = timed(lambda: [launch(s, i) for i in range(COP) for s in shapes], COP * 3)
print(f"interleaved: {t * 3:.2f} us per triple  vs  {tot:.2f} us sum of repeated", flush=True)
