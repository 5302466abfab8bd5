"""development: s_memtime stamps of the K-split-waves GEMM (GEMM_DIAG=6 build of gemm_ks.hip via LFAMD_HIP_SO)."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from llamafile_amd import sgemm, synth, _hip, ggml_types as T
m, k, n = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 4096, 512)
copies = int(sys.argv[4]) if len(sys.argv) > 4 else 8
PER = int(sys.argv[5]) if len(sys.argv) > 5 else 4  # stamps per stage (12 with -DKS_STAMP_KSTEPS)
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
print("rc", _hip.lib().lfamd_debug_ks_stamps(buf))
a = np.array(buf[:], dtype=np.int64).reshape(2, 256)
for kh in range(2):
    t = a[kh][a[kh] > 0]
    print(f"kh={kh}: n {len(t)} total cycles {t[-1] - t[0]}")
    d = np.diff(t)
    print("  prologue->stage0:", d[0])
    st = d[1:1 + PER * ((len(d) - 1) // PER)].reshape(-1, PER)  # per stage: [vmcnt wait, barrier wait, issue, k-loop(to next stage's first stamp)]
    for j, row in enumerate(st[:20]):
        if PER == 4:
            print(f"  stage {j:2d}: wait_vm {row[0]:5d}  barrier {row[1]:5d}  reads+issue {row[2]:5d}  ksteps {row[3]:5d}   sum {row.sum():5d}")
        else:
            print(f"  stage {j:2d}: wait_vm {row[0]:5d}  barrier {row[1]:5d}  reads+issue {row[2]:5d}  ksteps {row[3:11].tolist()} consts {row[11]:5d}  sum {row.sum():5d}")
    print("  tail:", d[1 + PER * ((len(d) - 1) // PER):].tolist())
