import sys, torch
sys.path.insert(0, ".")
from llamafile_amd import sgemm, ggml_types as T
sgemm.init(0)
for (m, k, n) in [(4096, 4096, 512), (14336, 4096, 512)]:
    for t, dt in ((T.F16, torch.float16), (T.BF16, torch.bfloat16)):
        raw = (torch.randn(m, k, device="cuda") * 0.05).to(dt).view(torch.uint8).view(m, k * 2)
        W = sgemm.upload_weights(t, raw, m, k)
        x = torch.randn(n, k, device="cuda")
        us = sgemm.time_mul_mat(W, x.view(torch.uint8).view(n, k * 4), T.F32, n, warmup=3, iters=20)
        print(T.NAMES[t], (m, k, n), f"{us:.1f} us  {2.0 * m * k * n / us / 1e6:.0f} TFLOP/s")
