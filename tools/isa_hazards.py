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


def check_decode_hygiene(path, extra_flags=()):
    """The decode GEMVs keep their weight prefetch only while hipcc can COUNT the loads in flight: a FLAT memory
    instruction anywhere in the kernel (a pointer that lost its address space) or a stack frame (closures that were not
    promoted to registers) makes its wait-count pass drain vmcnt(0) in front of every use — found the hard way in round 2.
    -> {kernel: [problems]} for every kernel of the translation unit."""
    flags = [f for f in FLAGS if f != "-fno-slp-vectorize"] + list(extra_flags)
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run([HIPCC, *flags, path, "-o", f.name], check=True, stderr=subprocess.DEVNULL)
        text = open(f.name).read()
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
    return res


if __name__ == "__main__":
    total = 0
    defs = tuple(a for a in sys.argv[1:] if a.startswith("-D"))
    for p in (a for a in sys.argv[1:] if not a.startswith("-D")):
        for k, bad in check_file(p, defs).items():
            total += len(bad)
            print(f"{p}: {k[:90]}: {len(bad)} hazard(s)")
            for l in bad[:5]:
                print("    ", l[:140])
    sys.exit(1 if total else 0)
