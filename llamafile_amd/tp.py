"""Tensor-parallel sharding of the mat-mul path.

The reference has no collective: its row-split mode shards weight rows and GATHERS the partial outputs on
a main GPU with peer copies (/root/reference/llama.cpp.patches/patches/ggml-cuda.cu.patch:17853-18153).
Here every rank owns a slice and the residual-stream partial sums are all-reduced:

  "rows"  (attn_q/k/v, ffn_gate/up): output features sharded -> each rank computes y[:, r0:r1], no comm.
  "cols"  (attn_output, ffn_down):   input features sharded at super-block granularity -> each rank computes a
          partial y over its k-slice; all_reduce(sum) gives y.  Quantisation is per block, so quantising a
          k-slice gives exactly the blocks of the full row.
  "vocab" (output.weight):           rows sharded, logits all-gathered.

Pure index arithmetic on raw GGUF tensors ([rows, row_bytes] uint8) — no compute here.
"""
from __future__ import annotations

import numpy as np

from . import ggml_types as T


def row_range(rows: int, rank: int, world: int) -> tuple[int, int]:
    if rows % world:
        raise ValueError(f"{rows} rows do not shard over {world} ranks")
    per = rows // world
    return rank * per, (rank + 1) * per


def col_range(cols: int, t: int, rank: int, world: int) -> tuple[int, int]:
    blk = T.BLCK[t]
    if cols % (world * blk):
        raise ValueError(f"k={cols} is not a multiple of world*block = {world}*{blk}: replicate this tensor instead")
    per = cols // world
    return rank * per, (rank + 1) * per


def shard_weight(raw: np.ndarray, t: int, rows: int, cols: int, mode: str, rank: int, world: int):
    """-> (raw_shard [rows_l, row_bytes_l], rows_l, cols_l)."""
    if world == 1:
        return raw, rows, cols
    if mode in ("rows", "vocab"):
        r0, r1 = row_range(rows, rank, world)
        return np.ascontiguousarray(raw[r0:r1]), r1 - r0, cols
    if mode == "cols":
        c0, c1 = col_range(cols, t, rank, world)
        b0, b1 = c0 // T.BLCK[t] * T.TYPE_SIZE[t], c1 // T.BLCK[t] * T.TYPE_SIZE[t]
        return np.ascontiguousarray(raw[:, b0:b1]), rows, c1 - c0
    raise ValueError(mode)


def shard_activation(x: np.ndarray, t: int, mode: str, rank: int, world: int) -> np.ndarray:
    """f32 activations [n, k]: only "cols" ops consume a k-slice."""
    if world == 1 or mode != "cols":
        return x
    c0, c1 = col_range(x.shape[1], t, rank, world)
    return np.ascontiguousarray(x[:, c0:c1])
