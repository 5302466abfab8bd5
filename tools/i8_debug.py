"""development: where does the int8 body differ from the exact-code f16 body?"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from llamafile_amd import sgemm, synth, _hip, ggml_types as T
m, k, n = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2048, 512, 512)
sgemm.init(0)
A = synth.random_weights(T.Q4_K, m, k, 41)
x = synth.random_activations(n, k, 42)
W = sgemm.upload_weights(T.Q4_K, A, m, k)
Bd = torch.from_numpy(x).cuda().view(torch.uint8).view(n, k * 4)
C1 = sgemm.mul_mat(W, Bd, T.F32, n=n).cpu().numpy()
C2 = sgemm.mul_mat(W, Bd, T.F32, n=n, flags=sgemm.host_variant_flags() | _hip.FLAG_PRECISE).cpu().numpy()
bad = np.abs(C1 - C2) > 1e-4 * np.abs(C2).max()
print("shape", m, k, n, "bad fraction", bad.mean())
tok, row = np.nonzero(bad)
if len(tok):
    print("tokens: min", tok.min(), "max", tok.max(), " tok%64 hist", np.bincount(tok % 64, minlength=64).tolist())
    print("rows: row%128 hist", np.bincount(row % 128, minlength=128).tolist())
    print("token tiles (tok//64) hist", np.bincount(tok // 64).tolist())
    print("row blocks (row//128) hist", np.bincount(row // 128).tolist())
    print("examples", [(int(t), int(r), float(C1[t, r]), float(C2[t, r])) for t, r in list(zip(tok, row))[:6]])
