"""development: s_memtime sums of the Q8_0 loader-wave batch body (tools/build_diag.sh gemm_lf -DLF_STAMPS=1, then LFAMD_HIP_SO): compute
wave 0 and loader wave 4 of one work-group; where a stage's time goes (barrier wait / K-steps; vmcnt wait / barrier wait / issue)."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, ".")
from llamafile_amd import sgemm, synth, _hip, ggml_types as T
m, k, n = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 4096, 512)
copies = int(sys.argv[4]) if len(sys.argv) > 4 else 8
sgemm.init(0)
raw = synth.random_weights_torch(T.Q8_0, m, k, seed=1)
Ws = [sgemm.upload_weights(T.Q8_0, raw, m, k) for _ in range(copies)]
x = torch.randn(n, k, device="cuda")
B = x.view(torch.uint8).view(n, k * 4)
for _ in range(2):
    for W in Ws:
        out = sgemm.mul_mat(W, B, T.F32, n=n)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
rc = _hip.lib().lfamd_debug_lf_stamps(buf)
a = [int(v) for v in buf]
nq = max(1, a[2])
print(f"{m} x {k} x {n}, LFAMD_LF_NT={os.environ.get('LFAMD_LF_NT', 'default')}: rc {rc}, {nq} stages (s_memtime ticks)")
print(f"  compute wave 0: barrier wait {a[0] / nq:7.1f}  K-steps {a[1] / nq:7.1f} per stage; start -> loop done {a[3]}, -> stored {a[4]}")
print(f"  loader wave 4 : vmcnt wait {a[8] / nq:7.1f}  barrier wait {a[9] / nq:7.1f}  issue {a[10] / nq:7.1f} per stage; start -> done {a[11]}")
