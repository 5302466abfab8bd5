"""How the default prefill numerics (scaled f16 operands on the MFMA body, one f16 rounding per operand) behave over a
DEPTH of layers: a 32-layer residual feed-forward chain (RMS-norm -> gate / up (Q4_K) -> silu * up -> down (Q4_K or Q6_K like
a Q4_K_M file) -> residual add), 512 tokens, then an output matrix (Q6_K, 32000 rows); every mat-mul through lfamd_mul_mat.
Run twice — default flags and LFAMD_FLAG_PRECISE (exact integer codes, f32 scales) — and compare the final logits.
Writes one JSON object (profiles/r02_chain_error.json)."""
import json, sys
import numpy as np, torch
sys.path.insert(0, ".")
from llamafile_amd import sgemm, synth, _hip, ggml_types as T
from llamafile_amd.llama_shapes import use_more_bits

sgemm.init(0)
D, FF, V, L, N = 4096, 14336, 32000, 32, 512
layers = []
for il in range(L):
    hi = T.Q6_K if use_more_bits(il, L) else T.Q4_K
    lay = {}
    for name, t, m, k in (("gate", T.Q4_K, FF, D), ("up", T.Q4_K, FF, D), ("down", hi, D, FF)):
        raw = synth.random_weights_torch(t, m, k, 5000 + 3 * il + len(name))
        lay[name] = sgemm.upload_weights(t, raw, m, k)
    layers.append(lay)
out_w = sgemm.upload_weights(T.Q6_K, synth.random_weights_torch(T.Q6_K, V, D, 4999), V, D)
g = torch.Generator(device="cuda"); g.manual_seed(11)
x0 = torch.randn((N, D), device="cuda", generator=g)

def norm(x):
    return x * torch.rsqrt((x * x).mean(dim=1, keepdim=True) + 1e-6)

def mm(W, x, flags):
    return sgemm.mul_mat(W, x.contiguous().view(torch.uint8).view(x.shape[0], -1), T.F32, flags=flags)

def run(flags, x_in=None):
    x = (x0 if x_in is None else x_in).clone()
    for lay in layers:
        h = norm(x)
        a, b = mm(lay["gate"], h, flags), mm(lay["up"], h, flags)
        y = mm(lay["down"], torch.nn.functional.silu(a) * b, flags)
        x = x + y * (0.5 / y.abs().mean().clamp_min(1e-9)) * x.abs().mean()  # keep the residual branch at half the stream's size
    return mm(out_w, norm(x), flags)

base = sgemm.host_variant_flags()
fast = run(base).double()
exact = run(base | _hip.FLAG_PRECISE).double()
d = (fast - exact).abs()
rms = exact.pow(2).mean().sqrt()
res = {"layers": L, "tokens": N, "shape": [D, FF, V],
       "normwise_max_abs_over_max_abs": float(d.max() / exact.abs().max()),
       "rms_err_over_rms": float(d.pow(2).mean().sqrt() / rms),
       "worst_abs_err_over_abs_plus_rms": float((d / (exact.abs() + rms)).max()),
       "fraction_beyond_1e-3_of_abs_plus_rms": float((d > 1e-3 * (exact.abs() + rms)).double().mean()),
       "argmax_agreement": float((fast.argmax(dim=1) == exact.argmax(dim=1)).double().mean())}
# the chain's own sensitivity: the EXACT path on an input perturbed by 3e-4 (relative, random) — the size of the error one
# scaled-operand mat-mul injects — tells how much of the drift above is amplification by this random network
noise = torch.randn(x0.shape, device="cuda", generator=g)
pert = run(base | _hip.FLAG_PRECISE, x0 * (1.0 + 3e-4 * noise)).double()
dp = (pert - exact).abs()
res["exact_path_input_perturbed_3e-4"] = {"rms_err_over_rms": float(dp.pow(2).mean().sqrt() / rms),
                                          "normwise": float(dp.max() / exact.abs().max()),
                                          "argmax_agreement": float((pert.argmax(dim=1) == exact.argmax(dim=1)).double().mean())}
pert2 = run(base | _hip.FLAG_PRECISE, x0 * (1.0 + 3e-6 * noise)).double()
dp2 = (pert2 - exact).abs()
res["exact_path_input_perturbed_3e-6"] = {"rms_err_over_rms": float(dp2.pow(2).mean().sqrt() / rms)}
print(json.dumps(res))
