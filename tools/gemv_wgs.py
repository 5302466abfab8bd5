"""development: per-work-group entry / exit times (s_memrealtime, 10 ns ticks) of the decode GEMV (GEMV_DIAG build via LFAMD_HIP_SO):
how long after the first work-group the others start (dispatch + XCD skew), when each ends, and the per-wave stamps of two of them."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from llamafile_amd import sgemm, synth, _hip, ggml_types as T
m, k = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 14336)
WT = getattr(T, sys.argv[3]) if len(sys.argv) > 3 else T.Q4_K
sgemm.init(0)
Ws = [sgemm.upload_weights(WT, synth.random_weights_torch(WT, m, k, seed=s), m, k) for s in range(12)]
x = torch.randn(1, k, device="cuda")
B = x.view(torch.uint8).view(1, k * 4)
for rep in range(3):
    for W in Ws:
        out = sgemm.mul_mat(W, B, T.F32, n=1)
torch.cuda.synchronize()
L = _hip.lib()
wg = (C.c_ulonglong * 2048)()
print("rc", L.lfamd_debug_gemv_wgs(wg))
a = np.array(wg[:], dtype=np.int64).reshape(512, 4)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
ent, ext = (a[:, 0] - t0) * 10, (a[:, 1] - t0) * 10  # ns
xcc = (a[:, 2] >> 32) & 0xf
print(f"{len(a)} work-groups; entry: min 0, median {np.median(ent):.0f}, p90 {np.percentile(ent, 90):.0f}, max {ent.max():.0f} ns")
print(f"exit: min {ext.min():.0f}, median {np.median(ext):.0f}, p90 {np.percentile(ext, 90):.0f}, max {ext.max():.0f} ns; life median {np.median(ext - ent):.0f} ns")
for xc in range(8):
    s = xcc == xc
    if s.any():
        print(f"  XCD {xc}: {s.sum():3d} wgs  entry {ent[s].min():5.0f}..{ent[s].max():5.0f}  exit {ext[s].min():5.0f}..{ext[s].max():5.0f}  life {np.median((ext - ent)[s]):5.0f}")
clk = a[:, 3] / np.maximum(1, (a[:, 1] - a[:, 0])) / 10.0  # cycles per ns = GHz
print(f"shader clock during the launch: median {np.median(clk) * 1000:.0f} MHz")
buf = (C.c_ulonglong * 1024)()
print("rc", L.lfamd_debug_gemv_stamps(buf))
s = np.array(buf[:512], dtype=np.int64).reshape(2, 16, 16)
for g in (0, 1):
    for w in (0, 5, 10, 15):
        t = s[g, w]
        t = t[t > 0]
        print("wg", g, "wave", w, "ns since first wg entry:", ((t - t0) * 10).tolist())
