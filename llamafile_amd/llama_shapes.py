"""Mat-mul inventories of the BASELINE.json model configs (shapes only — there are no model files
offline; weights are synthetic).

Each entry is one GGML_OP_MUL_MAT of a transformer layer: (name, ggml type, m rows, k cols,
input id, shard) where
  input id  names the activation the op consumes (ops with the same id share one quantised input,
            like attn_q/k/v do in the real graph), and
  shard     is how tensor parallelism splits it: "rows" = output features sharded,
            no communication; "cols" = input features sharded, partial sums all-reduced on the
            residual stream; "vocab" = rows sharded, logits all-gathered.
Q4_K_M recipe (upstream llama.cpp llama_tensor_get_type at the pinned commit, external knowledge —
SURVEY.md Appendix B): attn_v and ffn_down use Q6_K in layers where use_more_bits(i) holds, Q4_K
elsewhere; output.weight is Q6_K; everything else Q4_K.
"""
from __future__ import annotations

from dataclasses import dataclass

from . import ggml_types as T


@dataclass(frozen=True)
class MatMul:
    name: str
    type: int
    m: int
    k: int
    input: str
    shard: str


def use_more_bits(i_layer: int, n_layers: int) -> bool:
    return i_layer < n_layers // 8 or i_layer >= 7 * n_layers // 8 or (i_layer - n_layers // 8) % 3 == 2


def llama_q4_k_m(n_layers: int, n_embd: int, n_ff: int, n_head_kv_dim: int, n_vocab: int) -> list[list[MatMul]]:
    """Per-layer op lists + a final [output] list."""
    layers = []
    for il in range(n_layers):
        hi = T.Q6_K if use_more_bits(il, n_layers) else T.Q4_K
        layers.append([
            MatMul(f"blk.{il}.attn_q", T.Q4_K, n_embd, n_embd, "attn_in", "rows"),
            MatMul(f"blk.{il}.attn_k", T.Q4_K, n_head_kv_dim, n_embd, "attn_in", "rows"),
            MatMul(f"blk.{il}.attn_v", hi, n_head_kv_dim, n_embd, "attn_in", "rows"),
            MatMul(f"blk.{il}.attn_output", T.Q4_K, n_embd, n_embd, "attn_out_in", "cols"),
            MatMul(f"blk.{il}.ffn_gate", T.Q4_K, n_ff, n_embd, "ffn_in", "rows"),
            MatMul(f"blk.{il}.ffn_up", T.Q4_K, n_ff, n_embd, "ffn_in", "rows"),
            MatMul(f"blk.{il}.ffn_down", hi, n_embd, n_ff, "ffn_down_in", "cols"),
        ])
    layers.append([MatMul("output", T.Q6_K, n_vocab, n_embd, "out_in", "vocab")])
    return layers


def llama3_8b_q4_k_m() -> list[list[MatMul]]:
    return llama_q4_k_m(32, 4096, 14336, 1024, 128256)


def llama3_70b_q4_k_m() -> list[list[MatMul]]:
    return llama_q4_k_m(80, 8192, 28672, 1024, 128256)


def llama3_8b_q8_0() -> list[list[MatMul]]:
    out = []
    for layer in llama3_8b_q4_k_m():
        out.append([MatMul(o.name, T.Q8_0, o.m, o.k, o.input, o.shard) for o in layer])
    return out


def weight_bytes(op: MatMul) -> int:
    return op.m * T.row_size(op.type, op.k)


def gemv_algorithmic_bytes(op: MatMul, n: int = 1) -> int:
    """weights once + quantised activations + f32 output."""
    return weight_bytes(op) + n * T.row_size(T.VEC_DOT[op.type], op.k) + n * op.m * 4


def gemm_flops(op: MatMul, n: int) -> int:
    return 2 * op.m * op.k * n


# ---- the same inventory read from a model file (f-4: llama-bench / localscore run from a GGUF, localscore/benchmark.cpp:93-145)
_GGUF_OPS = (("attn_q", "attn_in", "rows"), ("attn_k", "attn_in", "rows"), ("attn_v", "attn_in", "rows"),
             ("attn_output", "attn_out_in", "cols"), ("ffn_gate", "ffn_in", "rows"), ("ffn_up", "ffn_in", "rows"),
             ("ffn_down", "ffn_down_in", "cols"))


def from_gguf(g) -> tuple[list[list[MatMul]], dict]:
    """-> (per-layer op lists + a final [output] list, {op name: GGUFTensor}) from the tensor directory of an open
    llamafile_amd.gguf.GGUFFile: every blk.N.{attn_q, attn_k, attn_v, attn_output, ffn_gate, ffn_up, ffn_down}.weight it holds
    (types and shapes as stored) and output.weight (token_embd.weight when the file ties the two).  Tensors whose type has no
    block format in this module (nbytes == 0) are an error: the bench would silently skip work."""
    by_name = {t.name: t for t in g.tensors}
    layers, tensors = [], {}
    il = 0
    while any(f"blk.{il}.{op}.weight" in by_name for op, _, _ in _GGUF_OPS):
        ops = []
        for op, inp, shard in _GGUF_OPS:
            t = by_name.get(f"blk.{il}.{op}.weight")
            if t is None:
                continue
            if not t.nbytes:
                raise ValueError(f"{t.name}: ggml type {t.type} has no block format here")
            name = f"blk.{il}.{op}"
            ops.append(MatMul(name, t.type, int(t.ne[1]), int(t.ne[0]), inp, shard))
            tensors[name] = t
        layers.append(ops)
        il += 1
    out = by_name.get("output.weight") or by_name.get("token_embd.weight")
    if out is not None and out.nbytes:
        layers.append([MatMul("output", out.type, int(out.ne[1]), int(out.ne[0]), "out_in", "vocab")])
        tensors["output"] = out
    if not layers:
        raise ValueError("no llama mat-mul tensors (blk.N.attn_q.weight ...) in this file")
    return layers, tensors
