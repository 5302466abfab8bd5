"""N > 1 path on CPU: two gloo ranks, each computing its tensor-parallel shard (the oracle stands in for the
GPU kernels), all-reduce / all-gather, compared with the unsharded result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, t, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from llamafile_amd import ggml_types as T, synth, tp
        from oracle import ora
        m, k, n = 64, 1024, 3
        A = synth.random_weights(t, m, k, 5)  # same seed on every rank: the "model"
        x = synth.random_activations(n, k, 6)
        bt = T.VEC_DOT[t]
        out = {}
        for mode in ("rows", "cols", "vocab"):
            Al, ml, kl = tp.shard_weight(A, t, m, k, mode, rank, world)
            xl = tp.shard_activation(x, t, mode, rank, world)
            ok, Cl = ora.sgemm(t, Al, bt, ora.quantize(bt, xl), ml, n, kl)
            assert ok == 1
            y = torch.from_numpy(Cl.copy())
            if mode == "cols":
                dist.all_reduce(y)  # residual-stream partial sums
                out[mode] = y.numpy()
            else:
                parts = [torch.empty_like(y) for _ in range(world)]
                dist.all_gather(parts, y)
                out[mode] = np.concatenate([p.numpy() for p in parts], axis=1)
        if rank == 0:
            ok, full = ora.sgemm(t, A, bt, ora.quantize(bt, x), m, n, k)
            q.put((out, full))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tname", ["Q4_K", "Q6_K", "Q8_0"])
def test_tensor_parallel_matches_unsharded(tname):
    from llamafile_amd import ggml_types as T
    t = T.BY_NAME[tname]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, t, q)) for r in range(2)]
    for p in procs:
        p.start()
    out, full = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if tname == "Q8_0":
        # tinyBLAS_Q0 picks plain vs Kahan summation per mnpack tile (tinyblas_cpu.h:794-931), and the tiling
        # depends on m: a row shard is a different (m, n) problem, so only the last bits may differ
        assert np.abs(out["rows"] - full).max() <= 2e-6 * np.abs(full).max()
        assert np.abs(out["vocab"] - full).max() <= 2e-6 * np.abs(full).max()
    else:
        # row / vocab shards reproduce the same outputs bit for bit (same per-row arithmetic)
        assert np.array_equal(out["rows"], full)
        assert np.array_equal(out["vocab"], full)
    # column shards: the k-sum is split in two f32 partials, so only the last bits may differ
    assert np.abs(out["cols"] - full).max() <= 2e-6 * np.abs(full).max()


def test_shard_ranges_reject_unshardable():
    from llamafile_amd import ggml_types as T, tp
    with pytest.raises(ValueError):
        tp.col_range(4096 + 256, T.Q4_K, 0, 8)
    assert tp.col_range(14336, T.Q4_K, 7, 8) == (12544, 14336)
    assert tp.row_range(128256, 3, 8) == (48096, 64128)
