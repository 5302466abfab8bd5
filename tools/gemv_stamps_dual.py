"""development: in-kernel stamps of the two-type decode launch (attn_q/k Q4_K + attn_v Q6_K); GEMV_DIAG build of gemv_dual."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from llamafile_amd import sgemm, synth, _hip, ggml_types as T
k = 4096
sgemm.init(0)
sets = []
for s in range(16):
    sets.append([sgemm.upload_weights(t, synth.random_weights_torch(t, m, k, seed=s * 3 + i), m, k)
                 for i, (t, m) in enumerate(((T.Q4_K, 4096), (T.Q4_K, 1024), (T.Q4_K if len(sys.argv) > 1 else T.Q6_K, 1024)))])
x = torch.randn(1, k, device="cuda")
B = x.view(torch.uint8).view(1, k * 4)
for Ws in sets:
    outs = sgemm.mul_mat_multi(Ws, B, T.F32, n=1)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 1024)()
print("rc", _hip.lib().lfamd_debug_gemv_stamps(buf))
a = np.array(buf[:512], dtype=np.int64).reshape(2, 16, 16)
t0 = a[a > 0].min()
for g in (0, 1):
    for w in (0, 3, 7, 11, 15):
        t = a[g, w]
        t = t[t > 0]
        print("wg", g, "wave", w, "ticks(10ns) since first stamp:", (t - t0).tolist())

wb = (C.c_ulonglong * 2048)()
print("rc", _hip.lib().lfamd_debug_gemv_wgs(wb))
w = np.array(wb[:], dtype=np.uint64).reshape(512, 4)
live = w[:, 0] > 0
t0 = w[live, 0].min()
print("wgs", int(live.sum()))
rows = []
for b in np.nonzero(live)[0]:
    hw, xcc = int(w[b, 2]) & 0xffffffff, int(w[b, 2]) >> 32
    rows.append((int(b), int(w[b, 0] - t0), int(w[b, 1] - t0), xcc & 0xf, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf))
print("block entry exit(10ns) xcc se sh cu")
for r in rows[::8] + rows[-8:]:
    print(*r)
import collections
place = collections.Counter((r[3], r[4], r[5], r[6]) for r in rows)
print("distinct CUs", len(place), "max WGs on one CU", max(place.values()))
ent = np.array([r[1] for r in rows]); ex = np.array([r[2] for r in rows])
clk = np.array([w[b, 3] / max(1, (w[b, 1] - w[b, 0])) * 100.0 for b in np.nonzero(live)[0]])  # MHz
print("shader clock MHz over the work-groups: min/median/max", int(clk.min()), int(np.median(clk)), int(clk.max()))
print("entry: min/median/max", ent.min(), int(np.median(ent)), ent.max(), " exit: min/median/max", ex.min(), int(np.median(ex)), ex.max())
