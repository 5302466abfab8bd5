"""development: print in-kernel s_memtime stamps of the wide GEMM (GEMM_DIAG=3 build via LFAMD_HIP_SO)."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from llamafile_amd import sgemm, synth, _hip, ggml_types as T
m, k, n = 14336, 4096, 512
sgemm.init(0)
raw = synth.random_weights_torch(T.Q4_K, m, k, seed=1)
W = sgemm.upload_weights(T.Q4_K, raw, m, k)
x = torch.randn(n, k, device="cuda")
B = x.view(torch.uint8).view(n, k * 4)
for _ in range(3):
    out = sgemm.mul_mat(W, B, T.F32, n=n)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 512)()
L = _hip.lib()
print("rc", L.lfamd_debug_wide_stamps(buf))
a = np.array(buf[:], dtype=np.int64).reshape(8, 64)
for w in (0, 3, 4, 7):
    t = a[w]
    t = t[t > 0]
    d = np.diff(t)
    print("wave", w, "n", len(t), "total", t[-1] - t[0])
    print("  deltas:", d[:40].tolist())
