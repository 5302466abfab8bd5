"""BASELINE.json config 4: Mixtral-8x7B Q4_K_M, the llamafile_mixmul / GGML_OP_MUL_MAT_ID expert path on one MI355X.

Decode pass = per layer attn_q/k/v (fused), attn_output, and the three MUL_MAT_ID ops (ffn_gate_exps, ffn_up_exps with the
token's activations shared by both chosen experts; ffn_down_exps with one activation row per expert), then output.weight.
Experts are picked INSIDE the GEMV kernels from the device-resident routing table (no host sync), so the whole pass is one
hipGraph.  Synthetic weights (26 GB packed), random top-2 routing per layer.  Used by bench.py (key "config4") and
tools/bench_mixtral.py.  Reference path: llamafile_mixmul / llamafile_mixmul_iqk (llamafile/sgemm.h:25-28, 76-77),
ggml_cuda_mul_mat_id (ggml-cuda.cu.patch:18499-18635)."""
import ctypes as C
import time

import torch

from . import _hip, ggml_types as T, sgemm, synth
from .llama_shapes import use_more_bits


def run(n_layers: int = 32, iters: int = 20, prefill: int = 512, dev=None) -> dict:
    E, TOPK, D, FF, KV, V = 8, 2, 4096, 14336, 1024, 32000
    dev = dev or torch.device("cuda", torch.cuda.current_device())
    L = _hip.lib()
    flags = sgemm.host_variant_flags()
    seed = [1]

    def W(t, m, k):
        seed[0] += 1
        return sgemm.upload_weights(t, synth.random_weights_torch(t, m, k, seed[0], dev), m, k, dev)

    def stack(t, m, k):  # experts back to back
        ws = [W(t, m, k) for _ in range(E)]
        data = torch.cat([w.data for w in ws])
        return data, ws[0].nbytes

    layers = []
    wbytes = active = 0
    for il in range(n_layers):
        hi = T.Q6_K if use_more_bits(il, n_layers) else T.Q4_K
        lay = {"q": W(T.Q4_K, D, D), "k": W(T.Q4_K, KV, D), "v": W(hi, KV, D), "o": W(T.Q4_K, D, D), "hi": hi}
        lay["gate"], gb = stack(T.Q4_K, FF, D)
        lay["up"], ub = stack(T.Q4_K, FF, D)
        lay["down"], db = stack(hi, D, FF)
        g = torch.Generator(device="cpu")
        g.manual_seed(il)
        lay["plan"] = torch.stack([torch.randperm(E, generator=g)[:TOPK]]).to(torch.int32).to(dev)  # [1][2]
        layers.append(lay)
        attn = lay["q"].nbytes + lay["k"].nbytes + lay["v"].nbytes + lay["o"].nbytes
        wbytes += attn + lay["gate"].numel() + lay["up"].numel() + lay["down"].numel()
        active += attn + TOPK * (gb + ub + db)
    out_w = W(T.Q6_K, V, D)
    wbytes += out_w.nbytes
    active += out_w.nbytes
    torch.cuda.synchronize()

    x = torch.rand((1, D), device=dev) * 2 - 1
    xf = torch.rand((TOPK, FF), device=dev) * 2 - 1  # one activation row per chosen expert for ffn_down
    ws = torch.empty(1 << 26, dtype=torch.uint8, device=dev)
    res_g = torch.empty((1, TOPK, FF), device=dev)
    res_u = torch.empty((1, TOPK, FF), device=dev)
    res_ptrs = (C.c_void_p * 2)(res_g.data_ptr(), res_u.data_ptr())
    res_d = torch.empty((1, TOPK, D), device=dev)

    def ptr(t):
        return C.c_void_p(t.data_ptr())

    def mm(Ws, xin):
        return sgemm.mul_mat_multi(Ws, xin.view(torch.uint8).view(xin.shape[0], -1), T.F32, n=1, flags=flags, workspace=ws)

    def moe(stackd, t, rows, cols, thought, tasks, plan, res):
        rc = L.lfamd_mul_mat_id(t, ptr(stackd), rows, cols, E, T.F32, ptr(thought), thought.stride(0) * 4, tasks, 1, ptr(plan), TOPK,
                                ptr(res), ptr(ws), ws.numel(), flags, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        _hip.check(rc, "mul_mat_id")

    def moe_gate_up(lay):  # ffn_gate_exps + ffn_up_exps: same activations, same routing -> one launch (2 tensors x 2 thinkers)
        wp = (C.c_void_p * 2)(lay["gate"].data_ptr(), lay["up"].data_ptr())
        rc = L.lfamd_mul_mat_id_multi(T.Q4_K, 2, wp, FF, D, E, T.F32, ptr(x), x.stride(0) * 4, 1, 1, ptr(lay["plan"]), TOPK, res_ptrs,
                                      ptr(ws), ws.numel(), flags, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        _hip.check(rc, "mul_mat_id_multi")

    def decode_pass():
        for lay in layers:
            mm([lay["q"], lay["k"], lay["v"]], x)  # (one launch: the K-quant pair {Q4_K, Q6_K} is fused)
            mm([lay["o"]], x)
            moe_gate_up(lay)
            moe(lay["down"], lay["hi"], D, FF, xf, TOPK, lay["plan"], res_d)
        mm([out_w], x)

    decode_pass()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g.capture_begin(capture_error_mode="thread_local")
        decode_pass()
        g.capture_end()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        g.replay()
    torch.cuda.synchronize()
    dec_ms = (time.perf_counter() - t0) / iters * 1e3

    # prefill: the three expert ops of one layer over `prefill` tokens (device-side routing, one grouped MFMA launch each)
    n = prefill
    xp = torch.rand((n, D), device=dev) * 2 - 1
    xq = sgemm.quantize_rows(T.Q8_K, xp)
    xfq = sgemm.quantize_rows(T.Q8_K, torch.rand((n * TOPK, FF), device=dev) * 2 - 1)
    gcpu = torch.Generator(device="cpu")
    gcpu.manual_seed(7)
    planp = torch.stack([torch.randperm(E, generator=gcpu)[:TOPK] for _ in range(n)]).to(torch.int32).to(dev)
    lay = layers[0]

    def prefill_layer_moe():
        sgemm.mul_mat_id(lay["gate"], T.Q4_K, FF, D, E, xq, T.Q8_K, 1, n, planp, TOPK, flags=flags)
        sgemm.mul_mat_id(lay["up"], T.Q4_K, FF, D, E, xq, T.Q8_K, 1, n, planp, TOPK, flags=flags)
        sgemm.mul_mat_id(lay["down"], lay["hi"], D, FF, E, xfq, T.Q8_K, TOPK, n, planp, TOPK, flags=flags)

    prefill_layer_moe()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        prefill_layer_moe()
    torch.cuda.synchronize()
    pf_moe_ms = (time.perf_counter() - t0) / 3 * 1e3
    return {
        "config": "Mixtral-8x7B Q4_K_M, MUL_MAT_ID expert path, 1 x MI355X, synthetic weights, matmul-only",
        "layers": n_layers, "weight_bytes": wbytes, "active_bytes_per_token": active,
        "decode_pass_ms": round(dec_ms, 4), "decode_tokens_per_s": round(1e3 / dec_ms, 1),
        "decode_active_GBps": round(active / (dec_ms * 1e-3) / 1e9, 1),
        "prefill_tokens": n, "prefill_moe_ms_per_layer": round(pf_moe_ms, 3),
        "prefill_moe_TFLOPs": round(2.0 * n * TOPK * (2 * FF * D + D * FF) / (pf_moe_ms * 1e-3) / 1e12, 1),
    }
