"""Two tensor-parallel ranks on ONE device (gloo carries the bootstrap; both processes use cuda:0): the HIP kernels on
their shard shapes, the exchange step through the C ABI's one-shot peer all-reduce (IPC-shared exchange blocks,
write-through stores, flags, rank-order sum + fused residual add), compared with the oracle on the unsharded problem.
Replaces the reference's main-GPU gather (ggml-cuda.cu.patch:17853-18153).  What one device cannot show — coherence
between two GPUs' caches over xGMI — is stated in DESIGN.md; the protocol uses system-scope (sc0 sc1) accesses only."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from llamafile_amd import ggml_types as T, sgemm, synth, tp
        torch.cuda.set_device(0)
        sgemm.init(0)
        comm = tp.Comm(rank, world, use_rccl=False, oneshot_bytes=64 * 1024)
        res = {}
        for t, m, k in [(T.Q4_K, 4096, 4096), (T.Q6_K, 4096, 14336), (T.Q4_K, 8192, 8192)]:
            A = synth.random_weights(t, m, k, 5)  # same seed on every rank: the "model"
            x = synth.random_activations(1, k, 6)
            resid = synth.random_activations(1, m, 7)
            Al, ml, kl = tp.shard_weight(A, t, m, k, "cols", rank, world)
            xl = tp.shard_activation(x, t, "cols", rank, world)
            W = sgemm.upload_weights(t, Al, ml, kl)
            part = sgemm.mul_mat(W, torch.from_numpy(xl).cuda().view(torch.uint8).view(1, kl * 4), T.F32, n=1)
            rd = torch.from_numpy(resid).cuda()
            outs = []
            for rep in range(3):  # consecutive calls alternate the two slots and advance the sequence number
                out = torch.empty_like(part)
                comm.allreduce_add(part, rd, out)
                outs.append(out)
            torch.cuda.synchronize()
            assert comm.check() == 0
            assert all(torch.equal(outs[0], o) for o in outs[1:])
            res[(t, m, k)] = (outs[0].cpu().numpy(), part.cpu().numpy())
        # a message too large for the slot and no RCCL communicator: refused, not silently wrong
        big = torch.zeros(1 << 20, device="cuda")
        try:
            comm.allreduce_add(big)
            refused = False
        except Exception:
            refused = True
        q.put((rank, res, refused))
        dist.all_reduce(torch.zeros(1))
        comm.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_hip_shards_and_oneshot_allreduce(gpu, oracle):
    import torch.multiprocessing as mp
    from llamafile_amd import ggml_types as T, synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, res, refused = q.get(timeout=300)
        got[rank] = res
        assert refused
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for key in got[0]:
        t, m, k = key
        out0, part0 = got[0][key]
        out1, part1 = got[1][key]
        assert np.array_equal(out0, out1), "ranks disagree"  # rank-order sum: bit-identical everywhere
        resid = synth.random_activations(1, m, 7)
        assert np.array_equal(out0, (resid + part0) + part1)  # residual + rank 0 + rank 1, in that order
        A = synth.random_weights(t, m, k, 5)
        x = synth.random_activations(1, k, 6)
        rows = np.arange(0, m, 61)
        ok, G = oracle.sgemm(t, np.ascontiguousarray(A[rows]), T.Q8_K, synth.quantize_activations(T.Q8_K, x), len(rows), 1, k, nth=4)
        assert ok == 1
        want = G + resid[:, rows]
        assert np.abs(out0[:, rows] - want).max() <= 2e-6 * np.abs(want).max()


def test_single_process_communicators_over_one_device(gpu):
    """lfamd_comm_init_all (the ncclCommInitAll shape SURVEY.md section 5.8 proposes: one host thread, a device per shard),
    rehearsed with three ranks on cuda:0: every rank's call is issued from the one thread on its own stream, the kernels meet
    in the shared exchange blocks, every rank gets the rank-order sum (+ its residual) bit-identically; consecutive calls
    alternate the slots; all-gather through the same protocol; a message past the slot without RCCL is refused."""
    import ctypes as C
    import torch
    from llamafile_amd import _hip
    lib = _hip.lib()
    world, count = 3, 4096
    comms = (C.c_void_p * world)()
    devs = (C.c_int * world)(0, 0, 0)
    assert lib.lfamd_comm_init_all(comms, world, devs, 64 * 1024) == 0, lib.lfamd_last_error()
    try:
        g = torch.Generator().manual_seed(3)
        parts = [torch.randn(count, generator=g).cuda() for _ in range(world)]
        resid = [torch.randn(count, generator=g).cuda() for _ in range(world)]
        streams = [torch.cuda.Stream() for _ in range(world)]
        torch.cuda.synchronize()
        want = []  # the kernel's order: residual, then the ranks' partials in rank order (f32)
        for r in range(world):
            w = resid[r].clone()
            for p in parts:
                w = w + p
            want.append(w)
        for rep in range(4):
            outs = [torch.empty(count, device="cuda") for _ in range(world)]
            for r in range(world):
                assert lib.lfamd_comm_allreduce_add_f32(comms[r], parts[r].data_ptr(), resid[r].data_ptr(), outs[r].data_ptr(), count,
                                                        streams[r].cuda_stream) == 0, lib.lfamd_last_error()
            torch.cuda.synchronize()
            for r in range(world):
                assert lib.lfamd_comm_check(comms[r]) == 0
                assert torch.equal(outs[r], want[r])
        gath = [torch.empty(world * count, device="cuda") for _ in range(world)]
        for r in range(world):
            assert lib.lfamd_comm_allgather(comms[r], parts[r].data_ptr(), gath[r].data_ptr(), count * 4, streams[r].cuda_stream) == 0
        torch.cuda.synchronize()
        for r in range(world):
            assert torch.equal(gath[r], torch.cat(parts))
        big = torch.zeros(1 << 20, device="cuda")
        assert lib.lfamd_comm_allreduce_sum_f32(comms[0], big.data_ptr(), big.numel(), None) != 0  # no RCCL over one device
    finally:
        for r in range(world):
            lib.lfamd_comm_destroy(comms[r])
