"""development: what does the vendor's tuned PLAIN f16/bf16 GEMM (torch.matmul -> hipBLASLt) reach at the prefill shapes?
A ceiling for a GEMM with no dequantisation at all (not used by the product; measurement only)."""
import torch, time
torch.manual_seed(0)
for dt in (torch.float16, torch.bfloat16):
    for (m, k, n) in [(4096, 4096, 512), (14336, 4096, 512), (4096, 14336, 512), (128256, 4096, 512), (8192, 8192, 8192)]:
        copies = max(1, min(32, int(600e6 // (m * k * 2))))
        Ws = [torch.randn(m, k, device="cuda", dtype=dt) for _ in range(copies)]
        x = torch.randn(n, k, device="cuda", dtype=dt)
        for W in Ws: y = x @ W.t()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for W in Ws: y = x @ W.t()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): g.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (10 * copies)
        print(f"{dt} m={m} k={k} n={n}: {us:.1f} us  {2.0*m*k*n/us/1e6:.0f} TFLOP/s", flush=True)
