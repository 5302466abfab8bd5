"""development: random-shape stress of the batch paths — default (int8 / scaled-operand / loader-wave bodies, every tile / split
variant) against the exact bodies (LFAMD_FLAG_PRECISE; F16 / BF16: the 128 x 128 wide body) and against a rerun."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from llamafile_amd import sgemm, synth, _hip, ggml_types as T

sgemm.init(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
worst = 0.0
for c in range(cases):
    t = [T.Q4_K, T.Q5_K, T.Q6_K, T.Q8_0, T.F16, T.BF16, T.Q4_K][int(rng.integers(7))]
    m = int(rng.choice([1, 7, 32, 33, 100, 128, 129, 500, 1024, 2000, 4096, 5000, 9000]))
    n = int(rng.choice([9, 10, 31, 32, 33, 64, 65, 100, 128, 129, 200, 257, 512, 700]))
    k = 256 * int(rng.choice([1, 2, 3, 4, 5, 8, 9, 16, 17, 24]))
    if t == T.Q8_0 and rng.integers(2):
        k += 128  # (rows of whole 128-weight quads that are not whole 256-groups)
    if m * n * k > 3e10:
        continue
    W = sgemm.upload_weights(t, synth.random_weights_torch(t, m, k, int(rng.integers(1 << 30))), m, k)
    x = (torch.rand((n, k), device="cuda") * 2 - 1) * float(10.0 ** rng.uniform(-3, 3))
    xb = x.view(torch.uint8).view(n, k * 4)
    a = sgemm.mul_mat(W, xb, T.F32)
    b = sgemm.mul_mat(W, xb, T.F32)
    e = sgemm.mul_mat(W, xb, T.F32, flags=sgemm.host_variant_flags() | (_hip.FLAG_GEMM_WIDE if t in (T.F16, T.BF16) else _hip.FLAG_PRECISE))
    torch.cuda.synchronize()
    assert torch.isfinite(a).all(), (T.NAMES[t], m, n, k)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32)), ("rerun differs", T.NAMES[t], m, n, k)
    err = float((a - e).abs().max() / e.abs().max().clamp_min(1e-30))
    worst = max(worst, err)
    assert err <= 1.5e-3, (T.NAMES[t], m, n, k, err)  # both sides round (Q6_K's exact body above 2048)
print("cases ok, worst scaled-vs-exact", worst)
