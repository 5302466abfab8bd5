"""Static check of the gfx950 ISA hipcc emits for the hand-written kernels: no instruction may touch a VGPR that an LDS read
(`ds_read*`) or a global / buffer load has been issued into before the `s_waitcnt` that covers that load.

Why: the kernels issue loads from inline asm and wait with counted `s_waitcnt lgkmcnt(N)` / `vmcnt(N)` asm statements; the
register allocator is free to put a register-to-register copy of an asm output between the two (it did, for a second set of
"next" variables in gemm_lw.hip: an intermittent read of data still in flight that no test caught reliably).  This walks
every kernel linearly (loads complete in issue order per counter) and reports such instructions.

    python tools/isa_hazards.py llamafile_amd/csrc/gemm_lw.hip [more .hip files]
"""
import re
import subprocess
import sys
import tempfile

HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-S", "--cuda-device-only"]
_RNG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def _regs(tok):
    out = set()
    for a, b, c in _RNG.findall(tok):
        if c:
            out.add(int(c))
        else:
            out.update(range(int(a), int(b) + 1))
    return out


def check_asm(text):
    """-> {kernel: [offending instruction lines]}"""
    res = {}
    for name, body in re.findall(r"^(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end", text, re.S | re.M):
        lds, vm = [], []  # outstanding operations per counter, oldest first: sets of destination VGPRs
        bad = []
        for line in body.split("\n"):
            l = line.strip()
            if not l or l[0] in ";." or l.endswith(":"):
                continue
            op = l.split()[0]
            ops = l[len(op):].split(",")
            pend = set().union(*lds, *vm) if (lds or vm) else set()
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", l)
                if m:
                    n = int(m.group(1))
                    lds = lds[len(lds) - n:] if n else []
                m = re.search(r"vmcnt\((\d+)\)", l)
                if m:
                    n = int(m.group(1))
                    vm = vm[len(vm) - n:] if n else []
                continue
            if op.startswith("s_") :
                if op.startswith("s_load") or op.startswith("s_buffer_load"):
                    pass  # scalar loads share lgkmcnt but return out of order: the compiler waits lgkmcnt(0) for them
                continue
            used = set()
            for t in ops:
                used |= _regs(t)
            is_ds = op.startswith("ds_")
            is_vm = op.startswith(("global_", "buffer_", "flat_", "scratch_"))
            if is_ds or is_vm:
                dst = _regs(ops[0]) if ("read" in op or "load" in op) and " lds" not in l and "_lds_" not in op else set()
                src = used - dst if dst else used
                if src & pend or dst & pend:
                    bad.append(l)
                (lds if is_ds else vm).append(dst)
                continue
            if used & pend:
                bad.append(l)
        res[name] = bad
    return res


_SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")


def _sregs(tok):
    out = set()
    for a, b, c in _SREG.findall(tok):
        if c:
            out.add(int(c))
        else:
            out.update(range(int(a), int(b) + 1))
    return out


def check_sgpr_vmem(text):
    """VALU-written SGPR (v_readfirstlane / v_readlane) read by a vector-memory instruction fewer than five wait states later:
    hipcc pads this itself EXCEPT when the memory instruction sits inside an asm statement (a stale base address; round 3: a
    memory fault in gemm_kr on some shapes).  -> {kernel: [offending instruction lines]}"""
    res = {}
    for name, body in re.findall(r"^(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end", text, re.S | re.M):
        recent = []  # (sgpr set, wait states since)
        bad = []
        for line in body.split("\n"):
            l = line.strip()
            if not l or l[0] in ";." or l.endswith(":"):
                continue
            op = l.split()[0]
            states = 1
            if op == "s_nop":
                states = int(l.split()[1]) + 1
            if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                used = _sregs(l)
                for regs, age in recent:
                    if regs & used and age < 5:
                        bad.append(l)
                        break
            recent = [(r, a + states) for r, a in recent if a + states < 8]
            if op.startswith(("v_readfirstlane", "v_readlane")):
                recent.append((_sregs(l.split(",")[0]), 0))
        res[name] = bad
    return res


_DPP = ("quad_perm:", "row_shl:", "row_shr:", "row_ror:", "wave_shl", "wave_shr", "wave_rol", "wave_ror", "row_mirror", "row_half_mirror",
        "row_bcast", "row_newbcast", "dpp8:")
_TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def check_valu_hazards(text):
    """Data hazards of the gfx940 family that need SOFTWARE wait states (an independent instruction or `s_nop` between producer
    and consumer; CDNA3 ISA guide section 4.5, LLVM GCNHazardRecognizer).  hipcc pads them in code it schedules itself, but it
    cannot see into an inline-asm statement: a VALU producer or consumer written in asm is on its own (gemv_impl.h:
    `v_permlane*_swap` pair; gemm_wide_impl.h: `v_readfirstlane` + `global_load` in asm).  Rules (wait states required):
      VALU writes a VGPR  -> v_permlane16/32_swap touching it: 2     VALU writes a VGPR -> DPP operand: 2
      VALU writes a VGPR  -> v_readlane / v_readfirstlane source: 1  transcendental result -> other VALU: 1
      VALU writes an SGPR -> v_readlane / v_writelane lane select: 4;  -> VALU reading it as a constant: 2;  -> VMEM address: 5
      VALU writes VCC     -> v_div_fmas: 4                            VALU writes EXEC  -> DPP: 5
      SALU writes M0      -> LDS-DMA load (`*_load_lds_*` / `lds` modifier): 1
      packed 16-bit VALU result (`v_pk_*_f16 / _i16 / _u16`, or a result written to a register half: `dst_sel:` / op_sel's destination bit)
                          -> any VALU or MFMA reading that VGPR: 1 (LLVM's dst_sel forwarding hazard; hipcc pads it — and assumes
                          it of every asm statement's outputs — but not BETWEEN two instructions of one asm statement: gemm_lf.hip's
                          dequantisation blocks keep a multiply one block behind the add it reads)
    The walk is linear (these windows are at most five instructions long).  -> {kernel: [message + instruction]}"""
    res = {}
    for name, body in re.findall(r"^(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end", text, re.S | re.M):
        recent = []  # (kind, vgprs, sgprs, flags, age): producers of the last few wait states
        bad = []
        for line in body.split("\n"):
            l = line.split(";")[0].strip()
            if not l or l[0] == "." or l.endswith(":"):
                continue
            op = l.split()[0]
            rest = l[len(op):]
            ops = [t.strip() for t in rest.split(",")]
            states = int(l.split()[1]) + 1 if op == "s_nop" else 1
            is_valu = op.startswith("v_") and not op.startswith(("v_mfma", "v_smfmac"))
            is_vmem = op.startswith(("global_", "buffer_", "flat_", "scratch_"))
            is_dpp = is_valu and any(d in l for d in _DPP)
            two_dst = op.startswith(("v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co", "v_subbrev_co", "v_div_scale", "v_mad_u64",
                                     "v_mad_i64"))  # (vdst, sdst / vcc carry-out, sources ...)
            srcs = ops[2 if two_dst else 1:]
            src_v = set().union(*[_regs(t) for t in srcs]) if srcs else set()
            src_s = set().union(*[_sregs(t) for t in srcs]) if srcs else set()
            all_v = src_v | _regs(ops[0]) if ops else set()

            def need(cond, n, what):
                for kind, vg, sg, fl, age in recent:
                    if age < n and cond(kind, vg, sg, fl):
                        bad.append(f"[{what}: {n} wait states, {age} given] {l}")
                        return

            if op.startswith("v_permlane") and "swap" in op:
                need(lambda k, vg, sg, fl: k == "valu" and vg & all_v, 2, "VALU-written VGPR -> permlane swap")
            if is_dpp:
                need(lambda k, vg, sg, fl: k == "valu" and vg & src_v, 2, "VALU-written VGPR -> DPP")
                need(lambda k, vg, sg, fl: k == "valu" and "exec" in fl, 5, "VALU-written EXEC -> DPP")
            if op.startswith(("v_readlane", "v_readfirstlane")):
                need(lambda k, vg, sg, fl: k == "valu" and vg & _regs(ops[1]), 1, "VALU-written VGPR -> readlane source")
            if op.startswith(("v_readlane", "v_writelane")) and len(ops) > 2:
                sel = _sregs(ops[2])
                need(lambda k, vg, sg, fl: k == "valu" and sg & sel, 4, "VALU-written SGPR -> lane select")
            elif is_valu and src_s:
                need(lambda k, vg, sg, fl: k == "valu" and sg & src_s, 2, "VALU-written SGPR -> VALU constant")
            if is_valu and not op.startswith(_TRANS):
                need(lambda k, vg, sg, fl: k == "trans" and vg & src_v, 1, "transcendental result -> VALU")
            if op.startswith("v_"):  # (MFMAs included)
                need(lambda k, vg, sg, fl: k == "valu" and "half" in fl and vg & src_v, 1, "packed / half-register result -> VALU")
            if op.startswith("v_div_fmas"):
                need(lambda k, vg, sg, fl: k == "valu" and "vcc" in fl, 4, "VALU-written VCC -> v_div_fmas")
            if is_vmem:
                used_s = _sregs(l)
                need(lambda k, vg, sg, fl: k == "valu" and sg & used_s, 5, "VALU-written SGPR -> VMEM")
                if "_lds_" in op or re.search(r"\blds\b", l):
                    need(lambda k, vg, sg, fl: k == "salu" and "m0" in fl, 1, "SALU-written M0 -> LDS-DMA")
            recent = [(k, vg, sg, fl, a + states) for k, vg, sg, fl, a in recent if a + states < 6]
            if is_valu:
                dst = ops[0] if ops else ""
                dst2 = ops[1] if len(ops) > 1 and two_dst else ""
                fl = set()
                if "vcc" in dst or "vcc" in dst2 or (op.startswith("v_cmp") and "_e64" not in op and not dst.startswith("s")):
                    fl.add("vcc")
                if op.startswith("v_cmpx") or dst.startswith("exec"):
                    fl.add("exec")
                m_sel = re.search(r"op_sel:\[([01,]+)\]", l)
                if (op.startswith("v_pk_") and re.search(r"_(f16|bf16|i16|u16)(_e64)?$", op)) or ("dst_sel:" in l and "dst_sel:DWORD" not in l) or (m_sel and not op.startswith("v_pk_") and m_sel.group(1).endswith("1") and m_sel.group(1).count(",") >= 2):
                    fl.add("half")
                recent.append(("trans" if op.startswith(_TRANS) else "valu", _regs(dst), _sregs(dst) | _sregs(dst2), fl, 0))
                if op.startswith(_TRANS):  # (a transcendental is a VALU producer for the other rules as well)
                    recent.append(("valu", _regs(dst), set(), set(), 0))
            elif op.startswith("s_") and not op.startswith(("s_nop", "s_waitcnt", "s_barrier", "s_cbranch", "s_branch", "s_setprio", "s_sleep")):
                if ops and ops[0] == "m0":
                    recent.append(("salu", set(), set(), {"m0"}, 0))
        res[name] = bad
    return res


def check_decode_hygiene(path, extra_flags=()):
    """The decode GEMVs keep their weight prefetch only while hipcc can COUNT the loads in flight: a FLAT memory
    instruction anywhere in the kernel (a pointer that lost its address space) or a stack frame (closures that were not
    promoted to registers) makes its wait-count pass drain vmcnt(0) in front of every use — found the hard way in round 2.
    -> {kernel: [problems]} for every kernel of the translation unit."""
    return decode_hygiene(shipped_asm(path, extra_flags))


def shipped_asm(path, extra_flags=()):
    """gfx950 assembly of a translation unit with the flags of the shipped build (no -fno-slp-vectorize)."""
    flags = [f for f in FLAGS if f != "-fno-slp-vectorize"] + list(extra_flags)
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run([HIPCC, *flags, path, "-o", f.name], check=True, stderr=subprocess.DEVNULL)
        return open(f.name).read()


def decode_hygiene(text):
    res = {}
    for name, body in re.findall(r"^(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end", text, re.S | re.M):
        probs = []
        n_flat = len(re.findall(r"^\s*flat_(load|store|atomic)", body, re.M))
        if n_flat:
            probs.append(f"{n_flat} FLAT memory instruction(s)")
        if re.search(r"^\s*scratch_(load|store)", body, re.M):
            probs.append("scratch (stack) traffic")
        res[name] = probs
    return res


def check_file(path, extra_flags=()):
    """extra_flags: e.g. ("-DKS_CHECK_NB=8",) for kernels whose stage loop is laid out out of execution order (the walk is
    linear): a fixed trip count turns the loop into straight-line code."""
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run([HIPCC, *FLAGS, *extra_flags, path, "-o", f.name], check=True, stderr=subprocess.DEVNULL)
        text = open(f.name).read()
    res = check_asm(text)
    for k, bad in check_sgpr_vmem(text).items():
        res[k] = res.get(k, []) + ["[VALU-written SGPR -> VMEM] " + b for b in bad]
    for k, bad in check_valu_hazards(text).items():
        res[k] = res.get(k, []) + [b for b in bad if "-> VMEM" not in b]  # (that rule is check_sgpr_vmem's)
    return res


def check_file_valu(path, extra_flags=()):
    """The VALU wait-state rules alone, on the SHIPPED build of a translation unit (no -fno-slp-vectorize)."""
    return check_valu_hazards(shipped_asm(path, extra_flags))


if __name__ == "__main__":
    total = 0
    defs = tuple(a for a in sys.argv[1:] if a.startswith(("-D", "-mllvm", "-amdgpu")))
    valu_only = "--valu" in sys.argv[1:]
    for p in (a for a in sys.argv[1:] if not a.startswith("-")):
        for k, bad in (check_file_valu if valu_only else check_file)(p, defs).items():
            total += len(bad)
            print(f"{p}: {k[:90]}: {len(bad)} hazard(s)")
            for l in bad[:5]:
                print("    ", l[:140])
    sys.exit(1 if total else 0)
