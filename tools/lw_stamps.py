"""development: s_memtime stamps of the loader-wave GEMM (GEMM_DIAG=4 build via LFAMD_HIP_SO)."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from llamafile_amd import sgemm, synth, _hip, ggml_types as T
m, k, n = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (14336, 4096, 512)
sgemm.init(0)
W = sgemm.upload_weights(T.Q4_K, synth.random_weights_torch(T.Q4_K, m, k, seed=1), m, k)
x = torch.randn(n, k, device="cuda")
B = x.view(torch.uint8).view(n, k * 4)
for _ in range(3):
    out = sgemm.mul_mat(W, B, T.F32, n=n)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
print("rc", _hip.lib().lfamd_debug_lw_stamps(buf))
a = np.array(buf[:], dtype=np.int64).reshape(2, 128)
print("memtime", a[0,126], "realtime", a[0,127], "clock MHz", a[0,126] / max(a[0,127],1) * 100)
print("store cycles", a[0,125])
a[0,125:] = 0
t0 = a[a > 0].min()
for r, name in ((0, "compute"), (1, "loader")):
    t = a[r][a[r] > 0]
    print(name, "n", len(t), "total", t[-1] - t[0])
    print("  rel:", (t[:26] - t0).tolist())
    print("  deltas:", np.diff(t)[:36].tolist())
