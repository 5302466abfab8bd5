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


class Comm:
    """The C-ABI communicator (include/lfamd_hip.h, "collectives") bootstrapped over an existing torch.distributed
    process group: RCCL inside the HIP module for any size (`use_rccl`), plus the one-shot peer all-reduce for messages of
    up to `oneshot_bytes` (0 = none).  torch.distributed only carries the 128-byte id and the 64-byte IPC handles."""

    def __init__(self, rank: int, world: int, use_rccl: bool = True, oneshot_bytes: int = 0, group=None):
        import ctypes as C

        import torch
        import torch.distributed as dist

        from . import _hip
        self.L = _hip.lib()
        self.rank, self.world = rank, world
        self.h = C.c_void_p()
        idbuf = None
        if use_rccl and world > 1:
            idb = (C.c_char * 128)()
            if rank == 0:
                _hip.check(self.L.lfamd_comm_unique_id(idb), "lfamd_comm_unique_id")
            idt = torch.frombuffer(bytearray(bytes(idb)), dtype=torch.uint8).clone()  # CPU tensors: the gloo side of the group
            dist.broadcast(idt, src=0, group=group)
            idbuf = C.create_string_buffer(bytes(idt.numpy().tobytes()), 128)
        _hip.check(self.L.lfamd_comm_init(C.byref(self.h), rank, world, idbuf), "lfamd_comm_init")
        self.block = None
        self.oneshot_error = None
        if oneshot_bytes and world > 1:
            nbytes = int(self.L.lfamd_oneshot_bytes(oneshot_bytes))
            # fine-grained (uncached) device memory from the module: peers must see these stores inside a running kernel.
            # Every rank takes part in BOTH exchanges whatever happens to it locally (a rank that raised between them would leave
            # the others waiting in a collective): a local failure is carried to the vote at the end instead.
            err = None
            hb = (C.c_char * 64)()
            try:
                self.block = C.c_void_p()
                _hip.check(self.L.lfamd_oneshot_alloc(C.byref(self.block), nbytes), "lfamd_oneshot_alloc")
                _hip.check(self.L.lfamd_oneshot_export(self.block, hb), "lfamd_oneshot_export")
            except Exception as e:  # noqa: BLE001
                err = e
            mine = torch.frombuffer(bytearray(bytes(hb)), dtype=torch.uint8).clone()
            handles = [torch.empty(64, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(handles, mine, group=group)
            if err is None:
                try:
                    allh = C.create_string_buffer(b"".join(h.numpy().tobytes() for h in handles), 64 * world)
                    _hip.check(self.L.lfamd_oneshot_attach(self.h, self.block, nbytes, allh, oneshot_bytes), "lfamd_oneshot_attach")
                except Exception as e:  # noqa: BLE001
                    err = e
            # the vote doubles as the CPU-side barrier: every rank's flag block is zeroed and mapped before the first all-reduce
            vote = torch.tensor([0.0 if err is None else 1.0])
            dist.all_reduce(vote, group=group)
            if float(vote.item()) != 0.0:
                self.oneshot_error = err if err is not None else RuntimeError("the one-shot exchange failed on another rank")
                self.close()  # (every rank raises here: the caller may build an RCCL-only communicator instead)
                raise RuntimeError(f"one-shot peer all-reduce unavailable on this node ({self.oneshot_error})")

    def allreduce_add(self, partial, residual=None, out=None):
        """out = residual + sum over ranks of partial (f32, on the current stream)."""
        import ctypes as C

        import torch

        from . import _hip
        out = partial if out is None else out
        rc = self.L.lfamd_comm_allreduce_add_f32(self.h, C.c_void_p(partial.data_ptr()),
                                                 C.c_void_p(residual.data_ptr()) if residual is not None else None,
                                                 C.c_void_p(out.data_ptr()), partial.numel(),
                                                 C.c_void_p(torch.cuda.current_stream().cuda_stream))
        _hip.check(rc, "lfamd_comm_allreduce_add_f32")
        return out

    def allgather(self, send, recv):
        import ctypes as C

        import torch

        from . import _hip
        rc = self.L.lfamd_comm_allgather(self.h, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()),
                                         send.numel() * send.element_size(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        _hip.check(rc, "lfamd_comm_allgather")
        return recv

    def check(self) -> int:
        return int(self.L.lfamd_comm_check(self.h))

    def clear_error(self):
        from . import _hip
        _hip.check(self.L.lfamd_comm_clear_error(self.h), "lfamd_comm_clear_error")

    def close(self):
        if self.h:
            self.L.lfamd_comm_destroy(self.h)
            self.h = None
        if self.block:
            self.L.lfamd_oneshot_free(self.block)
            self.block = None
