"""development: s_memtime stamps of the int8 GEMM body (GEMM_DIAG=7 build of gemm_i8.hip via LFAMD_HIP_SO): waves 0 (token tile 0)
and 4 (token tile 1) of one work-group; per stage [vmcnt wait, barrier, compute]."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from llamafile_amd import sgemm, synth, _hip, ggml_types as T
m, k, n = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 4096, 512)
copies = int(sys.argv[4]) if len(sys.argv) > 4 else 8
sgemm.init(0)
raw = synth.random_weights_torch(T.Q4_K, m, k, seed=1)
Ws = [sgemm.upload_weights(T.Q4_K, raw, m, k) for _ in range(copies)]
x = torch.randn(n, k, device="cuda")
B = x.view(torch.uint8).view(n, k * 4)
for _ in range(2):
    for W in Ws:
        out = sgemm.mul_mat(W, B, T.F32, n=n)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 512)()
print("rc", _hip.lib().lfamd_debug_i8_stamps(buf))
a = np.array(buf[:], dtype=np.int64).reshape(2, 256)
t = a[0][a[0] > 0]
print(f"compute wave 0: n {len(t)} total cycles {t[-1] - t[0]}")
d = np.diff(t)
print("  launch -> stage 0 landed:", d[0])
body = d[1:-2]
st = body[:2 * (len(body) // 2)].reshape(-1, 2)
for j, row in enumerate(st):
    print(f"  stage {j:2d}: barrier {row[0]:5d}  compute {row[1]:5d}   sum {row.sum():5d}")
print("  last compute, store:", d[-2:].tolist())
