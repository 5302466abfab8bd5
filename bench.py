#!/usr/bin/env python3
"""bench.py — prefill + decode tokens/s of the quantized mat-mul path, Llama-3-8B Q4_K_M shapes.

One "step" = one pass of the hot path over one batch of synthetic input: a 512-token prefill
(every GGML_OP_MUL_MAT of the model at n=512: f32 activations -> quantise -> MFMA GEMM) followed by
128 decode passes (the same 225 mat-muls at n=1: quantise -> wave-reduction GEMV).  Only the
mat-mul operators are in the step (attention / norm / rope belong to the rest of the graph and are
out of scope); weights are random blocks of the Q4_K_M tensor types, resident in HBM
in the packed layout before the timed region starts.

N GPUs: tensor parallel — q/k/v/gate/up sharded by output rows, attn_output /
ffn_down by input columns with an all-reduce (RCCL over xGMI) of the partial residual-stream sums,
output.weight by vocabulary rows with an all-gather of the logits.  The total work is fixed, so
scaling is "strong".

Prints ONE JSON line (rank 0).  `roofline` is the dominant kernel (the decode Q4_K GEMV) measured
live with HIP events; `cpu_baseline` is the CPU restatement (oracle, kind "port") on the host cores.
"""
from __future__ import annotations

import argparse
import datetime
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from llamafile_amd import _hip, ggml_types as T, llama_shapes as LS, sgemm, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense f16/bf16 MFMA


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--prefill", type=int, default=512)
    p.add_argument("--decode", type=int, default=128)
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--force-dist", action="store_true",
                   help="rehearsal: initialise the process group and issue the collectives even at world size 1")
    p.add_argument("--model", default="llama3-8b-q4_k_m", choices=["llama3-8b-q4_k_m", "llama3-8b-q8_0", "llama3-70b-q4_k_m"])
    p.add_argument("--gguf", default=None, metavar="PATH",
                   help="build the op list from a GGUF model file (tensor types and shapes as stored) and run on ITS weights, "
                        "uploaded straight from the read-only mapping, instead of --model's synthetic inventory")
    p.add_argument("--no-extra-configs", action="store_true",
                   help="skip the BASELINE config 3 (Llama-3-8B Q8_0) and config 4 (Mixtral-8x7B expert path) legs of the line")
    return p.parse_args()


class Op:
    __slots__ = ("spec", "m", "k", "W", "args")


class Runner:
    """Holds the sharded weights and pre-allocated buffers; runs one pass of all mat-muls at batch n."""

    def __init__(self, layers, rank, world, batches, dev, collectives=False, comm=None, tensors=None):
        self.L = _hip.lib()
        self.rank, self.world, self.dev = rank, world, dev
        self.collectives = collectives
        self.comm = comm  # llamafile_amd.tp.Comm (the C ABI's collectives); None: torch.distributed (LFAMD_COLLECTIVES=torch)
        self.flags = sgemm.host_variant_flags()
        self.layers = []
        seed = 0x5EED0000
        for layer in layers:
            ops = []
            for spec in layer:
                o = Op()
                o.spec = spec
                o.m, o.k = spec.m, spec.k
                if world > 1:
                    if spec.shard in ("rows", "vocab"):
                        assert spec.m % world == 0
                        o.m = spec.m // world
                    else:
                        assert spec.k % (world * T.BLCK[spec.type]) == 0, (spec.name, spec.k, world)
                        o.k = spec.k // world
                seed += 1
                if tensors is not None:  # a model file: this rank's shard of the mapped bytes ([rows, row_bytes] uint8)
                    from llamafile_amd import tp
                    import numpy as np
                    shard, _, _ = tp.shard_weight(tensors[spec.name].array(), spec.type, spec.m, spec.k, spec.shard, rank, world)
                    raw = torch.from_numpy(np.array(shard, copy=True) if world > 1 else np.asarray(shard).copy()).to(dev)
                else:
                    raw = synth.random_weights_torch(spec.type, o.m, o.k, seed * 131 + rank, dev)
                o.W = sgemm.upload_weights(spec.type, raw, o.m, o.k, dev)
                del raw
                ops.append(o)
            self.layers.append(ops)
        torch.cuda.synchronize()
        # buffers per batch size
        self.buf = {}
        for n in batches:
            b = {"x": {}, "xq": {}, "out": {}}
            ws = 16
            for ops in self.layers:
                for o in ops:
                    key = (o.spec.input, o.k)
                    if key not in b["x"]:
                        g = torch.Generator(device=dev)
                        g.manual_seed(1234 + len(b["x"]))
                        b["x"][key] = torch.rand((n, o.k), device=dev, generator=g) * 2 - 1
                        vdt = T.VEC_DOT[o.spec.type]
                        b["xq"][key] = torch.empty((n, T.row_size(vdt, o.k)), dtype=torch.uint8, device=dev)
                    ws = max(ws, sgemm.workspace_bytes(o.spec.type, o.m, o.k, n))
            b["ws"] = torch.empty(ws, dtype=torch.uint8, device=dev)
            if collectives:
                vm = [o for ops in self.layers for o in ops if o.spec.shard == "vocab"]
                if vm:
                    b["gather"] = [torch.empty((n, vm[0].m), dtype=torch.float32, device=dev) for _ in range(world)]
                    b["gather_flat"] = torch.empty((world, n, vm[0].m), dtype=torch.float32, device=dev)
            self.buf[n] = b

    def weight_bytes(self):
        return sum(o.W.nbytes for ops in self.layers for o in ops)

    def _groups(self, ops):
        """Consecutive ops of one layer that read the same activations with the same k: what a backend's graph_compute
        fuses into one call (lfamd_mul_mat_multi_types: one launch at decode for one type or for the K-quant pair
        {Q4_K, Q6_K} — attn_q/k/v of a Q4_K_M file; per type for batches)."""
        groups = []
        for o in ops:
            g = groups[-1] if groups else None
            if g and g[0].spec.input == o.spec.input and g[0].k == o.k and g[0].spec.shard == o.spec.shard and len(g) < 4 and \
                    (g[0].spec.type == o.spec.type or {g[0].spec.type, o.spec.type} <= {T.Q4_K, T.Q6_K}):
                g.append(o)
            else:
                groups.append([o])
        return groups

    def prepare(self, n):
        """Pre-build the ctypes argument arrays of every launch of a pass at batch n."""
        b = self.buf[n]
        calls = []
        for ops in self.layers:
            groups = self._groups(ops)
            gi = 0
            while gi < len(groups):
                g = groups[gi]
                o0 = g[0]
                x = b["x"][(o0.spec.input, o0.k)]
                cnt = len(g)
                outs = []
                for j, o in enumerate(g):
                    key = (o.m if o.spec.shard != "vocab" else -o.m, j)
                    if key not in b["out"]:
                        b["out"][key] = torch.empty((n, o.m), dtype=torch.float32, device=self.dev)
                    outs.append(b["out"][key])
                A_arr = (C.c_void_p * cnt)(*[o.W.data.data_ptr() for o in g])
                C_arr = (C.c_void_p * cnt)(*[t.data_ptr() for t in outs])
                m_arr = (C.c_long * cnt)(*[o.m for o in g])
                t_arr = (C.c_int * cnt)(*[o.spec.type for o in g])
                calls.append(("mm", g, x, outs, A_arr, C_arr, m_arr, t_arr))
                gi += 1
        b["calls"] = calls

    def run_pass(self, n, only_type=None):
        """Launch every GGML_OP_MUL_MAT of the model at batch n on the current stream: f32 activations in
        (quantisation is fused into the kernels), sibling ops sharing an input fused per layer.  `only_type`: restrict to the launches all of whose matrices are of one weight type, no collectives
        (roofline measurement).  Returns (launches, ops)."""
        L, b = self.L, self.buf[n]
        if "calls" not in b:
            self.prepare(n)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        ws, wsn = C.c_void_p(b["ws"].data_ptr()), b["ws"].numel()
        launches = nops = 0
        for kind, g, x, outs, A_arr, C_arr, m_arr, t_arr in b["calls"]:
            o0 = g[0]
            types = {o.spec.type for o in g}
            if only_type is not None and types != {only_type}:
                continue
            rc = L.lfamd_mul_mat_multi_types(len(g), t_arr, A_arr, m_arr, o0.k, T.F32, C.c_void_p(x.data_ptr()),
                                             x.stride(0) * 4, n, C_arr, m_arr, ws, wsn, self.flags, stream)
            if rc:
                _hip.check(rc, "mul_mat_multi_types " + o0.spec.name)
            launches += 1 if n <= 8 else len(g)
            nops += len(g)
            if only_type is None and self.collectives:
                for o, out in zip(g, outs):
                    if o.spec.shard == "cols":  # partial sums of the residual stream
                        if self.comm is not None:
                            self.comm.allreduce_add(out)
                        else:
                            torch.distributed.all_reduce(out)
                    elif o.spec.shard == "vocab":  # vocabulary-row shards of the logits
                        if self.comm is not None:
                            if self.comm.has_rccl or out.numel() * 4 <= self.comm.oneshot:
                                self.comm.allgather(out, b["gather_flat"])
                            # (a same-device rehearsal has no RCCL: shards beyond the one-shot slot are not gathered)
                        else:
                            torch.distributed.all_gather(b["gather"], out)
        return launches, nops


def staged_prefill(runner, n):
    """The prefill pass when the ops in front of the mat-muls are this module's fused producers (lfamd_rms_norm_quantize /
    lfamd_swiglu_quantize writing the batch bodies' staged images): every call group that accepts an image gets one (the producers
    themselves are not part of the pass: they stand where the graph's norm / SwiGLU nodes stand anyway).  Returns a closure that
    launches the pass and the number of groups per input format."""
    L, b = runner.L, runner.buf[n]
    if "calls" not in b:
        runner.prepare(n)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws, wsn = C.c_void_p(b["ws"].data_ptr()), b["ws"].numel()
    images, plan, counts = {}, [], {"scaled": 0, "int8": 0, "f32": 0}

    def image_for(x, k, kind):
        key = (x.data_ptr(), kind)
        if key not in images:
            size = L.lfamd_staged_scaled_size(k, n) if kind == "scaled" else L.lfamd_staged_q8k_size(k, n)
            img = torch.empty(size, dtype=torch.uint8, device=x.device)
            _hip.check(L.lfamd_rms_norm_quantize(C.c_void_p(x.data_ptr()), x.stride(0) * 4, C.c_void_p(0), C.c_float(1e-5), n, k,
                                                 _hip.TYPE_STAGED_SCALED if kind == "scaled" else _hip.TYPE_STAGED_Q8K,
                                                 C.c_void_p(img.data_ptr()), 0, C.c_void_p(0), 0, stream), "rms_norm_quantize (staged)")
            images[key] = img
        return images[key]

    for kind_, g, x, outs, A_arr, C_arr, m_arr, t_arr in b["calls"]:
        o0 = g[0]
        choice = ("f32", T.F32, x, x.stride(0) * 4)
        # (attn_output's rows come from the attention, which is not one of this module's producers: f32 rows, staged by the call)
        for kind, bt in (() if o0.spec.input == "attn_out_in" else (("scaled", _hip.TYPE_STAGED_SCALED), ("int8", _hip.TYPE_STAGED_Q8K))):
            img = image_for(x, o0.k, kind)
            rc = L.lfamd_mul_mat_multi_types(len(g), t_arr, A_arr, m_arr, o0.k, bt, C.c_void_p(img.data_ptr()), 0, n, C_arr, m_arr, ws, wsn,
                                             runner.flags, stream)
            if rc == 0:
                choice = (kind, bt, img, 0)
                break
        counts[choice[0]] += 1
        plan.append((len(g), t_arr, A_arr, m_arr, o0.k, choice[1], choice[2], choice[3], C_arr))
    torch.cuda.synchronize()

    def run():
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for cnt, t_arr, A_arr, m_arr, k, bt, src, brb, C_arr in plan:
            rc = L.lfamd_mul_mat_multi_types(cnt, t_arr, A_arr, m_arr, k, bt, C.c_void_p(src.data_ptr()), brb, n, C_arr, m_arr, ws, wsn,
                                             runner.flags, st)
            if rc:
                _hip.check(rc, "mul_mat_multi_types (staged pass)")
    return run, counts


def cpu_share():
    """Host threads this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU box gives a
    1-GPU job 16 of its 256 hardware threads; 128 OpenMP threads under that quota were throttled at random, which is
    what made round 1's baseline scatter by 8x)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(layers, prefill, decode):
    """Time the CPU restatement (oracle: scalar C + OpenMP over ith like sgemm_matmul_test.cpp:32-40, kind "port") on a
    bounded sample of the same workload, with pinned threads:
      * prefill: n = `prefill` DIRECTLY on attn_output (Q4_K 4096 x 4096), ffn_gate (Q4_K 14336 x 4096) and layer 0's
        ffn_down (Q6_K 4096 x 14336), median of 3 each; every other op of the model is priced at its type's measured
        FLOP rate;
      * decode: n = 1 on the seven ops of layer 0, median of 5 each, x layers, output.weight at its byte share.
    Reports the median-based tokens/s, the spread of the repeats, and `unstable` if any op's repeats differ by more than
    1.5x (then the measurement is repeated once before giving up)."""
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    try:
        from oracle import ora
        ora.build()
    except Exception as e:  # the oracle is test infrastructure; the bench line survives without it
        return {"value": None, "unit": "tokens/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    import statistics
    nth = min(cpu_share(), ora.lib().ora_max_threads())
    layer = layers[0]

    def timed(spec, n, reps):
        A = synth.random_weights(spec.type, spec.m, spec.k, 1)
        vdt = T.VEC_DOT[spec.type]
        Bq = synth.quantize_activations(vdt, synth.random_activations(n, spec.k, 2))
        ora.sgemm_openmp(spec.type, A, vdt, Bq, spec.m, min(n, 8), spec.k, nth)  # warm: page in the weights, wake the team
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            ora.sgemm_openmp(spec.type, A, vdt, Bq, spec.m, n, spec.k, nth)
            ts.append(time.perf_counter() - t0)
        return ts

    def measure():
        spreads, med_min = [], []
        by_name = {s.name.split(".")[-1]: s for s in layer}
        rate = {}  # weight type -> FLOP/s at n = prefill
        pf_ops = [by_name["attn_output"], by_name["ffn_gate"], by_name["ffn_down"]]
        fl, tm = {}, {}
        for spec in pf_ops:
            ts = timed(spec, prefill, 3)
            spreads.append(max(ts) / min(ts))
            med_min.append(statistics.median(ts) / min(ts))
            fl[spec.type] = fl.get(spec.type, 0.0) + 2.0 * spec.m * spec.k * prefill
            tm[spec.type] = tm.get(spec.type, 0.0) + statistics.median(ts)
        for t in fl:
            if tm[t] > 0:
                rate[t] = fl[t] / tm[t]
        for ops in layers:  # (a type layer 0 does not hold, e.g. output.weight's: priced like the first one measured)
            for s in ops:
                rate.setdefault(s.type, next(iter(rate.values())))
        t_prefill = sum(2.0 * s.m * s.k * prefill / rate[s.type] for ops in layers for s in ops)
        t_dec_layer = 0.0
        for spec in layer:
            ts = timed(spec, 1, 5)
            spreads.append(max(ts) / min(ts))
            med_min.append(statistics.median(ts) / min(ts))
            t_dec_layer += statistics.median(ts)
        lb = sum(LS.weight_bytes(s) for s in layer)
        t_decode_token = t_dec_layer * ((len(layers) - 1) + LS.weight_bytes(layers[-1][0]) / lb)
        return t_prefill, t_decode_token, max(spreads), max(med_min), rate

    # the value is built from per-op MEDIANS; it is reproducible when every op's median sits within 1.5x of its fastest
    # repeat (a single preempted repeat among five does not move a median; max / min is reported beside it)
    t_prefill, t_decode_token, spread, mm, rate = measure()
    unstable = mm > 1.5
    if unstable:
        print(f"bench.py: cpu_baseline medians are {mm:.2f}x their fastest repeat (> 1.5x): measuring again", file=sys.stderr)
        t_prefill, t_decode_token, spread, mm, rate = measure()
        unstable = mm > 1.5
        if unstable:
            print(f"bench.py: cpu_baseline STILL unstable (median {mm:.2f}x the fastest repeat): the value below is not "
                  f"reproducible", file=sys.stderr)
    total = t_prefill + decode * t_decode_token
    return {
        "value": round((prefill + decode) / total, 3), "unit": "tokens/s", "cores": nth, "kind": "port",
        "median_over_fastest_repeat": round(mm, 3), "max_over_min_repeat": round(spread, 3), "unstable": unstable,
        "sample": f"n={prefill} timed directly on layer 0's attn_output, ffn_gate and ffn_down ("
                  + ", ".join(f"{by.m}x{by.k} {T.NAMES[by.type]}" for by in (next(s for s in layer if s.name.endswith(nm)) for nm in ("attn_output", "ffn_gate", "ffn_down")))
                  + f"), median of 3; n=1 on layer 0's {len(layer)} mat-muls, median of 5; {nth} pinned threads "
                  f"(OMP_PROC_BIND=close, the cgroup's CPU share); other ops priced at their type's measured rate "
                  f"({', '.join(f'{T.NAMES[t]} {r / 1e9:.1f} GFLOP/s' for t, r in rate.items())}); "
                  f"decode {1.0 / t_decode_token:.2f} tok/s, prefill {prefill / t_prefill:.2f} tok/s",
    }


def kernel_source_hash():
    """sha256 over the kernel sources: a PMC traffic profile is quoted only for the sources it was measured on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "llamafile_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "llamafile_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def newest_traffic_profile():
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")):
        m = re.match(r"r(\d+)_v(\d+)_", os.path.basename(f)) or re.match(r"r(\d+)()_", os.path.basename(f))
        key = (int(m.group(1)), int(m.group(2) or 0)) if m else (0, 0)
        if best is None or key > best[0]:
            best = (key, f)
    return best[1] if best else None


def make_comm(rank, world, same_device=False):
    """The C-ABI communicator for the tensor-parallel run.  The one-shot peer all-reduce (decode-sized messages) is
    verified once against a known answer before it is trusted: on a mismatch or a lost peer the communicator is rebuilt
    with RCCL only.  `same_device`: a rehearsal with several ranks on one GPU (RCCL refuses that): one-shot for every size."""
    from llamafile_amd import tp
    oneshot = int(os.environ.get("LFAMD_ONESHOT_BYTES", str((32 << 20) if same_device else 65536)))
    use_rccl = not same_device and world > 1
    try:
        comm = tp.Comm(rank, world, use_rccl=use_rccl, oneshot_bytes=oneshot)
    except RuntimeError as e:  # (raised on EVERY rank when the one-shot exchange failed on any: tp.Comm votes)
        if "one-shot peer all-reduce unavailable" not in str(e) or not use_rccl:
            raise
        if rank == 0:
            print(f"bench.py: {e}: RCCL for every size", file=sys.stderr)
        oneshot = 0
        comm = tp.Comm(rank, world, use_rccl=True, oneshot_bytes=0)
    comm.mode = ("RCCL" if use_rccl else "no RCCL") + (f" + one-shot peer kernel <= {oneshot} B" if oneshot and world > 1 else "")
    comm.selftest = "not run"
    if oneshot and world > 1:
        # 32 calls with data that changes per call and per rank, the ranks deliberately out of step (rank r sleeps r ms before
        # every fourth call: a consumer that arrives early must WAIT for its peers, one that arrives late must still find
        # their slots intact), every element checked against the closed form
        idx = torch.arange(4096, device="cuda", dtype=torch.float32) * 1e-3
        y = torch.empty(4096, device="cuda")
        bad = 0
        for it in range(32):
            if it % 4 == 3:
                time.sleep(1e-3 * rank)
            x = torch.full((4096,), float((rank + 1) * (it + 1)), device="cuda") + idx
            comm.allreduce_add(x, None, y)
            torch.cuda.synchronize()
            want = (it + 1) * world * (world + 1) / 2 + world * idx
            if not bool(torch.allclose(y, want, rtol=1e-6, atol=1e-6)):
                bad += 1
        good = comm.check() == 0 and bad == 0
        flag = torch.tensor([1 if good else 0])
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        comm.selftest = f"passed 32/32 on every rank" if int(flag.item()) else f"FAILED ({bad}/32 wrong on rank {rank}, check {comm.check()})"
        if not int(flag.item()):
            if rank == 0:
                print("bench.py: one-shot peer all-reduce failed its self-test on this node: RCCL for every size", file=sys.stderr)
            comm.close()
            comm = tp.Comm(rank, world, use_rccl=True, oneshot_bytes=0)
            comm.mode = "RCCL (one-shot self-test failed)"
            comm.selftest = "one-shot FAILED -> RCCL only"
    if use_rccl or "self-test failed" in comm.mode:  # the RCCL path too, on a message beyond the one-shot slot
        big = torch.full((1 << 18,), float(rank + 1), device="cuda")
        comm.allreduce_add(big)
        torch.cuda.synchronize()
        if abs(float(big[0].item()) - world * (world + 1) / 2) > 1e-3 or abs(float(big[-1].item()) - world * (world + 1) / 2) > 1e-3:
            raise RuntimeError("RCCL all-reduce through the C ABI returned a wrong sum")
    comm.oneshot_eff = oneshot if (oneshot and world > 1 and "self-test failed" not in comm.mode) else 0
    rccl_on = use_rccl or "self-test failed" in comm.mode
    comm.describe = lambda: (
        (f"all-reduce <= {comm.oneshot_eff} B: one-shot peer kernel in fine-grained memory (self-test {comm.selftest}); " if comm.oneshot_eff else "")
        + ("larger all-reduces and the logits all-gather: ncclAllReduce / ncclAllGather (RCCL, known-answer check passed)" if rccl_on
           else "no RCCL (ranks share a device): every size on the one-shot kernel")
        + ", through the C ABI")
    comm.has_rccl = use_rccl or "self-test failed" in comm.mode
    comm.oneshot = oneshot if "self-test failed" not in comm.mode else 0
    return comm


def config3_q8_0(a, dev):
    """BASELINE config 3: Llama-3-8B Q8_0 — the batch-1 decode GEMV (bit-exact tinyBLAS_Q0 restatement) and the prefill
    pass, 8.5 GB of packed weights, hipGraph replays timed with events."""
    layers = LS.llama3_8b_q8_0()
    r = Runner(layers, 0, 1, (a.prefill, 1), dev)
    resident = sum(o.W.resident_bytes for ops in r.layers for o in ops)
    res = {"model": "llama3-8b-q8_0", "weight_bytes": resident,  # what the tensors occupy in HBM (the images the kernels read)
           "gguf_bytes": r.weight_bytes(),  # the tensors' size in the file: what the decode roofline below counts
           "resident_images": "P80 only: the bit-exact vecdot GEMV and the f16 MFMA batch body read the same image (the file's 1.0625 B per weight)" +
                              (" + f16(d*q) rows for hipBLASLt (LFAMD_USE_BLASLT=1)" if _hip.lib().lfamd_vendor_gemm_available() else ""),
           "batch_gemm": "hipBLASLt (opt-in)" if _hip.lib().lfamd_vendor_gemm_available() else
                         "prep_lf_kernel + gemm_lf_q80_kernel (csrc/gemm_lf.hip: f16(d*q) built in registers from P80, loader waves, <= 1e-3)"}
    for n, reps, key in ((1, 20, "decode"), (a.prefill, 2, "prefill")):
        r.run_pass(n)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            g.capture_begin(capture_error_mode="thread_local")
            r.run_pass(n)
            g.capture_end()
        torch.cuda.current_stream().wait_stream(side)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res[f"{key}_pass_ms"] = round(ms, 4)
        res[f"{key}_tokens_per_s"] = round(n / (ms * 1e-3), 1)
        if n == 1:
            res["decode_GBps"] = round(r.weight_bytes() / (ms * 1e-3) / 1e9, 1)
            res["decode_hbm_frac"] = round(r.weight_bytes() / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        else:
            fl = sum(2.0 * o.m * o.k * n for ops in r.layers for o in ops)
            res["prefill_TFLOPs"] = round(fl / (ms * 1e-3) / 1e12, 1)
        del g
    del r
    return res


def spawn_ranks(n):
    """`python3 bench.py --gpus N` invoked bare (no launcher, WORLD_SIZE unset): this process touches no GPU; it starts N fresh
    child ranks of this script (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment — what
    torch.distributed.run would have set), relays rank 0's JSON line and fails if any rank fails."""
    import socket
    import subprocess

    have = torch.cuda.device_count()  # (counts devices without initialising one)
    shared = os.environ.get("LFAMD_DIST_BACKEND", "nccl") != "nccl"  # gloo rehearsal: the ranks may share device 0
    if have == 0:
        print(f"bench.py: --gpus {n} needs {n} MI355X devices; this machine shows none (torch.cuda.device_count() == 0)", file=sys.stderr)
        return 1
    if have < n and not shared:
        print(f"bench.py: --gpus {n} needs {n} devices, {have} visible (for a rehearsal of the ranks on one device set "
              f"LFAMD_DIST_BACKEND=gloo)", file=sys.stderr)
        return 1
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    line = procs[0].stdout.read()
    codes = [p.wait() for p in procs]
    if line:
        os.write(1, line)
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return next(c for _, c in bad) if all(c > 0 for _, c in bad) else 1
    return 0


def main():
    a0 = parse()
    if a0.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a0.gpus))
    # ONE JSON line on stdout: anything else a library prints there from C++ (gloo: "[Gloo] Rank 0 is connected ...") goes
    # to stderr — file descriptor 1 is pointed at stderr for the run and the line is written to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        line = run()
    finally:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
    if line is not None:
        os.write(real_stdout, (line + "\n").encode())
        if '"invalid":' in line:  # collectives failed their check after the timed region: the number is void
            sys.exit(3)


def run():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and a.gpus > 1:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or leave WORLD_SIZE unset: the "
              f"script then starts its own ranks)", file=sys.stderr)
        sys.exit(2)
    # one rank per GPU (the driver's launch); for a rehearsal on a 1-GPU box ranks may share device 0 with
    # LFAMD_DIST_BACKEND=gloo (RCCL refuses two ranks on one device)
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    sgemm.init(local)
    dist_on = world > 1 or a.force_dist
    comm = None
    abi_collectives = os.environ.get("LFAMD_COLLECTIVES", "abi") != "torch"
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("LFAMD_DIST_BACKEND", "nccl")
        if abi_collectives:
            # compute AND collectives through the C ABI (include/lfamd_hip.h): RCCL inside the HIP module + the one-shot
            # peer all-reduce for the decode-sized messages.  torch.distributed carries the bootstrap bytes, the barrier and
            # the max over the ranks' clocks on gloo; its own NCCL backend (created lazily, only if ever used) is the fallback
            # should the module's communicator fail its self-test on this node.
            same_device = backend != "nccl"
            # (ten minutes, not the default thirty, before a collective a lost rank never joins gives up)
            torch.distributed.init_process_group("gloo" if same_device else "cpu:gloo,cuda:nccl", rank=rank, world_size=world,
                                                 timeout=datetime.timedelta(seconds=600))
            try:
                comm = make_comm(rank, world, same_device=same_device)
            except Exception as e:  # noqa: BLE001
                print(f"bench.py[rank {rank}]: C-ABI communicator unavailable ({type(e).__name__}: {str(e)[:160]}); "
                      f"torch.distributed collectives instead", file=sys.stderr)
                comm = None
            ok = torch.tensor([1 if comm is not None else 0])
            torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)  # all ranks take the same path
            if not int(ok.item()) and comm is not None:
                comm.close()
                comm = None
        elif backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            torch.distributed.init_process_group(backend, rank=rank, world_size=world)

    gguf_file, gguf_tensors = None, None
    if a.gguf:
        from llamafile_amd import gguf as gguf_mod
        gguf_file = gguf_mod.GGUFFile(a.gguf)
        layers, gguf_tensors = LS.from_gguf(gguf_file)
        a.model = "gguf:" + os.path.basename(a.gguf)
    else:
        layers = {"llama3-8b-q4_k_m": LS.llama3_8b_q4_k_m, "llama3-8b-q8_0": LS.llama3_8b_q8_0,
                  "llama3-70b-q4_k_m": LS.llama3_70b_q4_k_m}[a.model]()
    runner = Runner(layers, rank, world, (a.prefill, 1), dev, collectives=dist_on, comm=comm, tensors=gguf_tensors)

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            if abi_collectives:
                torch.distributed.all_reduce(torch.zeros(1))  # (gloo, CPU tensor: no NCCL communicator is created for it)
            else:
                torch.distributed.barrier()
        torch.cuda.synchronize()

    # hipGraph capture of a whole pass (kernels + RCCL collectives).  If capturing the collectives is not
    # possible on this stack, fall back to eager launches rather than fail.
    # (gloo rehearsals cannot capture their host-side collectives, and a failed capture leaves gloo unusable: eager)
    use_graph = not a.no_graph and not (dist_on and comm is None and os.environ.get("LFAMD_DIST_BACKEND", "nccl") != "nccl")
    graphs = {}
    dead = []  # failed capture objects are kept alive: destroying a half-captured graph can crash the runtime
    if use_graph:
        cap_stream = torch.cuda.Stream(device=dev)

        def capture(n, only_type=None):
            # manual begin/end on a side stream (not torch.cuda.graph: its __exit__ raises from capture_end() BEFORE it
            # restores the current stream, which leaves every later launch on an invalidated capture stream)
            g = torch.cuda.CUDAGraph()
            cap_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cap_stream):
                g.capture_begin(capture_error_mode="thread_local")
                try:
                    runner.run_pass(n, only_type=only_type)
                except BaseException:
                    dead.append(g)
                    try:
                        g.capture_end()
                    except Exception:  # noqa: BLE001
                        pass
                    raise
                g.capture_end()
            torch.cuda.current_stream().wait_stream(cap_stream)
            return g

        try:
            for n in (a.prefill, 1):
                runner.run_pass(n)  # warm (also sets any kernel attributes before capture)
                barrier()
                graphs[n] = capture(n)
            barrier()
            for n in (a.prefill, 1):
                graphs[n].replay()
            barrier()
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                print(f"bench.py: graph capture unavailable ({type(e).__name__}: {str(e)[:200]}); running eagerly", file=sys.stderr)
            dead.extend(graphs.values())
            graphs = {}
            use_graph = False
            try:
                torch.cuda.synchronize()
            except Exception:  # noqa: BLE001
                pass

    def one_pass(n):
        if use_graph:
            graphs[n].replay()
        else:
            runner.run_pass(n)

    def step():
        one_pass(a.prefill)
        for _ in range(a.decode):
            one_pass(1)

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if abi_collectives else dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1000.0 / a.steps

    # ---- the collectives of the timed region are verified AFTER it: no peer was lost (lfamd_comm_check) and one more
    # known-answer all-reduce is right on every rank; a line measured over void sums is marked invalid and the run fails
    comm_invalid = None
    allreduce_us = None
    if dist_on and comm is not None and world > 1:
        ck = comm.check()
        xv = torch.full((4096,), float(rank + 1), device=dev)
        yv = torch.empty_like(xv)
        comm.allreduce_add(xv, None, yv)
        torch.cuda.synchronize()
        okv = ck == 0 and comm.check() == 0 and bool(torch.allclose(yv, torch.full_like(yv, world * (world + 1) / 2), rtol=1e-6))
        flag = torch.tensor([1 if okv else 0])
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        if not int(flag.item()):
            comm_invalid = f"collective verification after the timed region failed (rank {rank}: check {ck}, sum ok {okv})"
            print("bench.py: " + comm_invalid, file=sys.stderr)
        # what the residual-stream exchange of one decode pass costs: its 2 x n_layers all-reduces back to back
        n_ar = sum(1 for ops in runner.layers for o in ops if o.spec.shard == "cols")
        if n_ar:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            buf = torch.zeros(next(o.m for ops in runner.layers for o in ops if o.spec.shard == "cols"), device=dev)
            barrier()
            e0.record()
            for _ in range(n_ar):
                comm.allreduce_add(buf)
            e1.record()
            torch.cuda.synchronize()
            allreduce_us = round(e0.elapsed_time(e1) * 1e3, 1)

    # ---- phase split (informational): prefill pass and decode pass timed apart with events
    def time_region(fn, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        n_launch = 0
        for _ in range(reps):
            r = fn()
            n_launch += r[0] if isinstance(r, tuple) else (r or 0)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3, n_launch  # microseconds

    pf_us, _ = time_region(lambda: one_pass(a.prefill), 3)
    dc_us, _ = time_region(lambda: one_pass(1), 20)
    pf_us /= 3
    dc_us /= 20
    # the same prefill pass with LFAMD_FLAG_PRECISE: the exact integer-code MFMA bodies for Q4_K / Q5_K / Q6_K batches (2e-6 of
    # the oracle) instead of the scaled-operand ones (<= 1e-3) — what a bit-faithful prefill costs (eager launches, not in `value`)
    saved_flags = runner.flags
    runner.flags |= _hip.FLAG_PRECISE
    try:
        runner.run_pass(a.prefill)
        pfx_us, _ = time_region(lambda: runner.run_pass(a.prefill), 2)
        pfx_us /= 2
    finally:
        runner.flags = saved_flags

    # the same prefill pass behind fused producers (eager launches like the leg above; the f32 pass timed the same way beside it)
    staged_leg = None
    if world == 1 and not a.gguf:
        try:
            run_staged, counts = staged_prefill(runner, a.prefill)
            run_staged()
            st_us, _ = time_region(run_staged, 3)
            runner.run_pass(a.prefill)
            ef_us, _ = time_region(lambda: runner.run_pass(a.prefill), 3)
            staged_leg = {"pass_ms": round(st_us / 3e3, 3), "f32_rows_pass_ms_same_timing": round(ef_us / 3e3, 3), "call_groups": counts,
                          "what": "every call group whose body reads a staged image takes it from lfamd_rms_norm_quantize / "
                                  "lfamd_swiglu_quantize (LFAMD_TYPE_STAGED_SCALED / _Q8K) instead of staging f32 rows itself; the "
                                  "producers stand where the graph's norm / SwiGLU nodes stand and are not timed; same bits"}
        except Exception as e:  # noqa: BLE001
            print(f"bench.py: staged prefill leg skipped ({type(e).__name__}: {str(e)[:200]})", file=sys.stderr)

    # ---- roofline of the dominant kernel: the decode GEMV of the dominant weight type, all its
    # launches of one decode pass, back to back on the stream, timed with HIP events
    dom_type = T.Q8_0 if a.model == "llama3-8b-q8_0" else T.Q4_K
    if a.gguf:  # the type that holds most of the file's mat-mul bytes among launches of one type
        by_type = {}
        for c in (runner.buf[1].get("calls") or (runner.prepare(1), runner.buf[1]["calls"])[1]):
            ts = {o.spec.type for o in c[1]}
            if len(ts) == 1:
                by_type[next(iter(ts))] = by_type.get(next(iter(ts)), 0) + sum(o.W.nbytes for o in c[1])
        dom_type = max(by_type, key=by_type.get) if by_type else T.Q4_K
    dom_ops = [o for ops in runner.layers for g in runner._groups(ops) if {q.spec.type for q in g} == {dom_type} for o in g]
    launches_per_pass, _ = runner.run_pass(1, only_type=dom_type)
    dom_graph = None
    if use_graph:  # replayed like the passes above: eager launches leave host-side gaps between the kernels
        try:
            dom_graph = capture(1, only_type=dom_type)
        except Exception:  # noqa: BLE001
            dom_graph = None
    if dom_graph is not None:
        dom_graph.replay()
        us, _ = time_region(lambda: dom_graph.replay(), 20)
        n_launch = 20 * launches_per_pass
    else:
        us, n_launch = time_region(lambda: runner.run_pass(1, only_type=dom_type), 10)
    avg_us = us / n_launch
    # algorithmic bytes per launch: weights once + f32 activations (once per launch: sibling
    # ops fused into a launch share them) + f32 outputs
    dom_calls = [c for c in runner.buf[1]["calls"] if {o.spec.type for o in c[1]} == {dom_type}]
    alg_bytes = 0
    for kind, g, *_ in dom_calls:
        alg_bytes += sum(o.m * T.row_size(dom_type, o.k) + o.m * 4 for o in g) + g[0].k * 4
    avg_bytes = alg_bytes / launches_per_pass
    achieved = avg_bytes / (avg_us * 1e-6) / 1e9
    # HBM traffic per launch: from the rocprofv3 PMC passes of this same workload (tools/profile_round.sh:
    # FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM), stored
    # under profiles/ — counters cannot be read from inside this process
    traffic, traffic_src, tj = None, None, None
    tfile = newest_traffic_profile()
    if tfile and dom_type == T.Q4_K and world == 1 and a.model == "llama3-8b-q4_k_m":  # (the profile is of THIS workload)
        try:
            tj = json.load(open(tfile))
            rel = os.path.relpath(tfile, ROOT)
            if tj.get("kernel_source_sha256") == kernel_source_hash():
                traffic = tj["gemv_q4k"]["hbm_bytes_per_launch"]
                traffic_src = f"{rel} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH x2; same kernel sources)"
            else:  # counters cannot be read from inside this process: a profile of OTHER kernel sources is not this code's traffic
                traffic_src = f"{rel} was measured on different kernel sources (hash mismatch): traffic not reported"
                tj = None
        except (KeyError, ValueError, OSError):
            tj = None
    if dom_type == T.Q4_K:
        kname = ("gemv_kq_kernel<q4k_traits, 1, F32, {16 | 8 waves}, {1,2}>"
                 " (all decode launches of a pass whose matrices are all Q4_K)")
    elif dom_type == T.Q8_0:
        kname = "gemv_q80_kernel<1, F32, mode>"
    else:
        kname = f"decode GEMV launches whose matrices are all {T.NAMES[dom_type]}"
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": kname, "launches_per_pass": launches_per_pass, "mat_muls_per_pass": len(dom_ops),
                "avg_launch_us": round(avg_us, 3), "algorithmic_bytes_per_launch": int(avg_bytes)}

    # ---- secondary: the prefill GEMM at the north-star shape (4096 x 4096 x 512, 1 GPU shapes only)
    roofline_gemm = None
    if world == 1 and a.model == "llama3-8b-q4_k_m":
        o = runner.layers[0][0]
        xin = runner.buf[a.prefill]["x"][(o.spec.input, o.k)]
        gus = sgemm.time_mul_mat(o.W, xin.view(torch.uint8), T.F32, a.prefill, warmup=5, iters=50)
        fl = 2.0 * o.m * o.k * a.prefill
        tf = fl / (gus * 1e-6) / 1e12
        roofline_gemm = {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tf / MFMA_F16_PEAK_TFLOPS, 4),
                         "kernel": ("prep_i8_kernel + gemm_i8_kernel (128x64 tile, int8 matrix cores: exact integer sub-block dots, loader waves + one "
                                    "computing wave per SIMD; 128 tiles of 128x128 cannot fill 256 CUs)"
                                    if _hip.lib().lfamd_mul_mat_is_exact(o.W.type, o.m, o.k, a.prefill, runner.flags) else
                                    "prep_scaled_kernel + gemm_ks_kernel<Q4_K> (128x64 tile, K-split waves, scaled operands)"),
                         "shape": [o.m, o.k, a.prefill], "avg_launch_us": round(gus, 2)}
        # the same mat-mul when a fused producer (lfamd_rms_norm_quantize / lfamd_swiglu_quantize with LFAMD_TYPE_STAGED_Q8K) has
        # written the body's staged image: no staging launch in front of it (same bits; tests/test_gpu_norm_quant.py)
        L_ = _hip.lib()
        if L_.lfamd_mul_mat_takes_staged(o.W.type, o.m, o.k, a.prefill, runner.flags):
            image = torch.empty(L_.lfamd_staged_q8k_size(o.k, a.prefill), dtype=torch.uint8, device=dev)
            _hip.check(L_.lfamd_rms_norm_quantize(C.c_void_p(xin.data_ptr()), xin.stride(0) * 4, C.c_void_p(0), C.c_float(1e-5), a.prefill, o.k,
                                                  _hip.TYPE_STAGED_Q8K, C.c_void_p(image.data_ptr()), 0, C.c_void_p(0), 0,
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rms_norm_quantize (staged)")
            sus = sgemm.time_mul_mat(o.W, image, _hip.TYPE_STAGED_Q8K, a.prefill, warmup=5, iters=50)
            stf = fl / (sus * 1e-6) / 1e12
            roofline_gemm["from_producer_staged_image"] = {
                "avg_launch_us": round(sus, 2), "achieved": round(stf, 1), "frac": round(stf / MFMA_F16_PEAK_TFLOPS, 4),
                "what": "gemm_i8_kernel alone: the activations arrive as the staged image a fused producer wrote "
                        "(Btype = LFAMD_TYPE_STAGED_Q8K); the headline figure above is the GGML_OP_MUL_MAT boundary (f32 rows in, "
                        "staging launch included)"}
        # the same measurement on the largest Q4_K shape of the model (256x128 row-split body, gemm_kr.hip)
        big = [q for q in runner.layers[0] if q.W.type == T.Q4_K and q.m >= 8192]
        if big:
            q = big[0]
            xin = runner.buf[a.prefill]["x"][(q.spec.input, q.k)]
            gus = sgemm.time_mul_mat(q.W, xin.view(torch.uint8), T.F32, a.prefill, warmup=3, iters=20)
            tf = 2.0 * q.m * q.k * a.prefill / (gus * 1e-6) / 1e12
            roofline_gemm["large_shape"] = {"shape": [q.m, q.k, a.prefill], "avg_launch_us": round(gus, 2),
                                            "achieved": round(tf, 1), "frac": round(tf / MFMA_F16_PEAK_TFLOPS, 4),
                                            "kernel": "prep_scaled_kernel + gemm_kr_kernel<Q4_K> (256x128 tile, row-split waves, scaled operands)"}
            if L_.lfamd_mul_mat_takes_staged_scaled(q.W.type, q.m, q.k, a.prefill, runner.flags):  # (the same, behind a fused producer)
                image = torch.empty(L_.lfamd_staged_scaled_size(q.k, a.prefill), dtype=torch.uint8, device=dev)
                _hip.check(L_.lfamd_rms_norm_quantize(C.c_void_p(xin.data_ptr()), xin.stride(0) * 4, C.c_void_p(0), C.c_float(1e-5), a.prefill, q.k,
                                                      _hip.TYPE_STAGED_SCALED, C.c_void_p(image.data_ptr()), 0, C.c_void_p(0), 0,
                                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rms_norm_quantize (scaled image)")
                sus = sgemm.time_mul_mat(q.W, image, _hip.TYPE_STAGED_SCALED, a.prefill, warmup=3, iters=20)
                stf = 2.0 * q.m * q.k * a.prefill / (sus * 1e-6) / 1e12
                roofline_gemm["large_shape"]["from_producer_staged_image"] = {"avg_launch_us": round(sus, 2), "achieved": round(stf, 1),
                                                                              "frac": round(stf / MFMA_F16_PEAK_TFLOPS, 4)}

    # ---- small batches (a few sequences decoding together): one call of 8 tokens on the three decode shapes of a layer
    small_batch = None
    if world == 1 and a.model == "llama3-8b-q4_k_m" and not a.gguf:
        small_batch = {"tokens": 8, "kernel": "sb_prep_kernel + gemm_sb*_kernel (csrc/gemm_sb.hip: exact codes, one MFMA tile of 32 token slots)",
                       "timing": "back-to-back calls on ONE weight tensor (the small shapes stay in the Infinity Cache); figures with the "
                                 "weights streamed from HBM: profiles/r03_small_batch.txt, r03_small_batch16.txt"}
        for name in ("attn_output", "ffn_gate", "ffn_down"):
            ops8 = [q for q in runner.layers[0] if q.spec.name.endswith(name + ".weight") or q.spec.name.endswith(name)]
            if not ops8:
                continue
            q = ops8[0]
            x8 = torch.rand((8, q.k), device=dev) * 2 - 1
            us8 = sgemm.time_mul_mat(q.W, x8.view(torch.uint8).view(8, q.k * 4), T.F32, 8, warmup=3, iters=30)
            us1 = sgemm.time_mul_mat(q.W, x8[:1].contiguous().view(torch.uint8).view(1, q.k * 4), T.F32, 1, warmup=3, iters=30)
            small_batch[name] = {"shape": [q.m, q.k], "type": T.NAMES[q.W.type], "us_8_tokens": round(us8, 2), "us_1_token": round(us1, 2)}

    if roofline_gemm and tj:
        # HBM traffic of the GEMM tiles: averages over the launches of the profiled bench (each tile serves several
        # shapes of the model), from the same PMC passes as the GEMV's
        roofline_gemm["traffic_avg_over_model_launches"] = {k: v.get("hbm_bytes_per_launch") for k, v in tj.items()
                                                            if isinstance(v, dict) and k.startswith("gemm")}

    if rank != 0:
        if dist_on:
            torch.distributed.destroy_process_group()
        return None

    tokens = a.prefill + a.decode
    out = {
        "metric": "prefill + decode tokens/sec, Llama-3-8B Q4_K_M, 1/2/4/8 MI355X vs CPU tinyBLAS",
        "value": round(tokens / (ms_per_step / 1000.0), 2),
        "unit": "tokens/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "int8",
        "data": "synthetic" if not a.gguf else f"weights of {os.path.basename(a.gguf)} (GGUF v{gguf_file.version}, {len(gguf_tensors)} mat-mul tensors), synthetic activations",
        "config": {
            "workload": f"{a.model} mat-muls ({sum(len(l) for l in layers)} GGML_OP_MUL_MAT per pass, f32 activations in, quantisation fused, sibling ops fused at decode" + "), "
                        f"{a.prefill}-token prefill + {a.decode} decode, matmul-only",
            "numerics": "decode: exact int8 x int4/int6 block dot products with f32 scales (v_dot4_i32_i8); prefill: Q4_K on grids of up "
                        "to 256 tiles of 128x128 (attn_q and its Q4_K siblings, attn_output, ffn_down) on the int8 matrix cores, exact integer sub-block dots "
                        "(2e-6 of the oracle); larger Q4_K grids and Q6_K on f16 MFMA with scaled operands f16(d*sc*q) x f16(d8*code) "
                        "(<= 1e-3 relative, measured ~3e-4; exact integer codes with LFAMD_FLAG_PRECISE)",
            "model": a.model, "prefill_tokens": a.prefill, "decode_tokens": a.decode,
            "parallelism": "single GPU" if world == 1 else f"tp{world} (all-reduce on attn_output/ffn_down: {comm.describe() if comm is not None else 'torch.distributed ' + os.environ.get('LFAMD_DIST_BACKEND', 'nccl')})",
            "hip_graph": use_graph, "weight_bytes_per_gpu": runner.weight_bytes(),
            "prefill_tokens_per_s": round(a.prefill / (pf_us * 1e-6), 1),
            "decode_tokens_per_s": round(1.0 / (dc_us * 1e-6), 1),
            "prefill_pass_ms": round(pf_us / 1e3, 3), "decode_pass_ms": round(dc_us / 1e3, 4),
            "prefill_exact": {"flags": "LFAMD_FLAG_PRECISE", "pass_ms": round(pfx_us / 1e3, 3),
                              "tokens_per_s": round(a.prefill / (pfx_us * 1e-6), 1), "numerics": "2e-6 of the oracle"},
            **({"prefill_behind_fused_producers": staged_leg} if staged_leg else {}),
            "decode_launches_per_pass": sum(1 for _ in runner.buf[1]["calls"]),
            "decode_GBps_whole_pass": round(runner.weight_bytes() / (dc_us * 1e-6) / 1e9, 1),
        },
        "roofline": roofline,
    }
    if allreduce_us is not None:
        out["roofline"]["allreduce_us"] = allreduce_us  # the 2 x n_layers residual-stream all-reduces of one decode pass, back to back
    if comm_invalid:
        out["invalid"] = comm_invalid
    if roofline_gemm:
        out["roofline_prefill_gemm"] = roofline_gemm
    if small_batch:
        out["small_batch"] = small_batch
    # ---- BASELINE configs 3 and 4 beside the headline config, so the driver's record carries them (1 GPU, default model)
    if world == 1 and a.model == "llama3-8b-q4_k_m" and not a.no_extra_configs:
        del runner, graphs
        torch.cuda.empty_cache()
        out["config3"] = config3_q8_0(a, dev)
        torch.cuda.empty_cache()
        try:
            from llamafile_amd import mixtral_bench
            out["config4"] = mixtral_bench.run(32, 60, a.prefill, dev)
        except Exception as e:  # noqa: BLE001 (the headline line must survive a failure of a side leg)
            out["config4"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
        torch.cuda.empty_cache()
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(layers, a.prefill, a.decode)
    else:
        out["cpu_baseline"] = None
    if dist_on:
        torch.distributed.destroy_process_group()
    return json.dumps(out)


if __name__ == "__main__":
    main()
