"""development: in-kernel s_memrealtime stamps (10 ns ticks) of the decode GEMV (GEMV_DIAG build via LFAMD_HIP_SO)."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from llamafile_amd import sgemm, synth, _hip, ggml_types as T
m, k = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 4096)
WT = getattr(T, sys.argv[3]) if len(sys.argv) > 3 else T.Q4_K
sgemm.init(0)
Ws = [sgemm.upload_weights(WT, synth.random_weights_torch(WT, m, k, seed=s), m, k) for s in range(24)]
x = torch.randn(1, k, device="cuda")
B = x.view(torch.uint8).view(1, k * 4)
for W in Ws:
    out = sgemm.mul_mat(W, B, T.F32, n=1)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 1024)()
print("rc", _hip.lib().lfamd_debug_gemv_stamps(buf))
a = np.array(buf[:512], dtype=np.int64).reshape(2, 16, 16)
t0 = a[a > 0].min()
for g in (0, 1):
    for w in range(16):
        t = a[g, w]
        t = t[t > 0]
        print("wg", g, "wave", w, "ticks(10ns) since first stamp:", (t - t0).tolist())
