#!/usr/bin/env python3
"""Registers, spills and LDS of every kernel in the built libmpcore.so, read from the code object's metadata (no
compile, no GPU): objcopy the .hip_fatbin section, unbundle the gfx950 code object, llvm-readelf --notes.
The register screen kernels sit at the 128-VGPR edge of four wavefronts per SIMD; a change elsewhere in the file
can tip them into spilling (it did once) -- tests/test_abi_and_host.py asserts they do not.
Usage: python scripts/kernel_resources.py [substring]"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "matching-pursuit_amd", "lib", "libmpcore.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_resources(lib=LIB):
    """-> {mangled kernel name: dict(vgpr, spill, sgpr, lds, scratch)}"""
    with tempfile.TemporaryDirectory() as tmp:
        fat, dev = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", lib, os.path.join(tmp, "x")])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={dev}"])
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", dev], text=True)
    out = {}
    keys = {".vgpr_count": "vgpr", ".vgpr_spill_count": "spill", ".sgpr_count": "sgpr",
            ".group_segment_fixed_size": "lds", ".private_segment_fixed_size": "scratch"}
    # one YAML list item per kernel under amdhsa.kernels (items start at two spaces of indentation; the kernel's
    # arguments are nested lists further in)
    for rec in re.split(r"(?m)^  - (?=\.)", notes)[1:]:
        cur = {}
        for line in rec.splitlines():
            m = re.match(r"\s{0,4}(\.[a-z_]+):\s+(\S+)", line)
            if m and m.group(1) in keys:
                cur[keys[m.group(1)]] = int(m.group(2))
            elif m and m.group(1) == ".name":
                cur["name"] = m.group(2)
        if "name" in cur:
            out[cur.pop("name")] = cur
    return out


def _device_disassembly(lib=LIB):
    with tempfile.TemporaryDirectory() as tmp:
        fat, dev = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", lib, os.path.join(tmp, "x")])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={dev}"])
        return subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", dev], text=True)


_DIS = {}


def kernel_instructions(mangled_substring, lib=LIB):
    """-> [(address, mnemonic, operands)] of the first kernel whose mangled name contains the substring."""
    if lib not in _DIS:
        _DIS[lib] = _device_disassembly(lib).split("\n")
    lines = _DIS[lib]
    start = next(i for i, l in enumerate(lines) if mangled_substring in l and l.endswith(">:"))
    out = []
    for l in lines[start + 1:]:
        if re.match(r"^[0-9a-f]+ <.*>:$", l):
            break
        m = re.match(r"^\s+(\S+)\s*(.*?)//\s*([0-9A-F]+):", l)
        if m:
            out.append((int(m.group(3), 16), m.group(1), m.group(2).strip()))
    return out


def loops(insts):
    """Backward branches of a kernel -> [(first index, last index)] of the loop bodies."""
    addr = {a: i for i, (a, _, _) in enumerate(insts)}
    out = []
    for i, (a, op, args) in enumerate(insts):
        if op.startswith("s_cbranch") or op == "s_branch":
            off = int(args.split()[0])
            if off >= 32768:
                off -= 65536
            nxt = insts[i + 1][0] if i + 1 < len(insts) else a + 4
            if off < 0 and nxt + off * 4 in addr:
                out.append((addr[nxt + off * 4], i))
    return out


# mangled-name substrings of the kernels whose pair loop bench.py prices: (template arguments as they mangle)
SCREEN_KERNELS = {
    ("persistent", 10): "fft_persistent_kernelILi10ELi4", ("persistent", 11): "fft_persistent_kernelILi11ELi2",
    ("persistent", 12): "fft_persistent_kernelILi12ELi1",
    ("screen", 10): "fft_screen_kernelILi10ELb0", ("screen", 11): "fft_screen_kernelILi11ELb0",
    ("screen", 12): "fft_screen_kernelILi12ELb0", ("screen", 13): "fft_screen_kernelILi13ELb0",
    ("screen", 14): "fft_screen_kernelILi14ELb0",
}


def pair_loop_range(insts):
    """(first, last) instruction index of the largest loop that loads exactly one transform's pair spectrum -- sixteen complex
    points per thread: sixteen 8-byte loads, or eight 16-byte ones where the 2048-point transform takes adjacent first-pass
    columns -- and holds no matrix instruction; None if there is none."""
    best = None
    for lo, hi in loops(insts):
        body = insts[lo:hi + 1]
        if sum(2 if op == "global_load_dwordx4" else 1 for _, op, _ in body if op.startswith("global_load")) != 16 \
                or any("mfma" in op for _, op, _ in body):
            continue
        if best is None or hi - lo > best[1] - best[0]:
            best = (lo, hi)
    return best


def screen_pair_loop(kind, log_m, lib=LIB):
    """The screen's loop over atom pairs in the built code object: one trip = one M-point transform of X * P by the
    workgroup's threads (16 points per thread), spectrum product, transform and running maxima.  It is the largest loop
    whose body holds exactly the pair-spectrum loads of one transform (sixteen points per thread) and no matrix instruction.
    -> dict(valu, packed, lds, barriers, instructions) per trip and thread."""
    insts = kernel_instructions(SCREEN_KERNELS[(kind, log_m)], lib)
    best = pair_loop_range(insts)
    if best is None:
        raise RuntimeError(f"no pair loop found in {SCREEN_KERNELS[(kind, log_m)]}")
    body = insts[best[0]:best[1] + 1]
    return dict(valu=sum(op.startswith("v_") for _, op, _ in body), packed=sum(op.startswith("v_pk_") for _, op, _ in body),
                lds=sum(op.startswith("ds_") for _, op, _ in body), barriers=sum(op == "s_barrier" for _, op, _ in body),
                instructions=len(body))


def _vregs(operand):
    """VGPR numbers named by one operand (v7, v[4:5]; modifiers like neg_lo:[..] or offset:.. name none)."""
    m = re.fullmatch(r"v(\d+)", operand)
    if m:
        return [int(m.group(1))]
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


def _sources(op, ops):
    """The operands an instruction READS (vector registers): all of them for stores, all but the first otherwise."""
    if op.startswith(("ds_write", "global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "ds_add",
                      "ds_max", "ds_min", "ds_or", "ds_and")) and "_rtn" not in op:
        return ops
    return ops[1:]


def lds_read_hazards(insts, reads=("ds_read_b64",)):
    """LDS reads whose destination is CONSUMED before the counter says it has arrived.

    The hardware does not interlock on LDS returns: an instruction that reads the destination of a ds_read before an
    s_waitcnt has brought lgkmcnt down far enough sees the register's OLD content.  The compiler places those waits for
    the reads it emits itself; csrc/mpfft.inc's transform issues its reads as `asm volatile("ds_read_b64 ...")` and
    waits with `asm volatile("s_waitcnt lgkmcnt(0)")` -- an order the compiler cannot see (a
    multiply once moved in front of the wait: wrong picks in 31 of 60 encodes, no marker).  For every read of the named
    kinds this walks forward through its straight-line code (up to the next branch) with the counter's semantics -- LDS
    operations return in order, `s_waitcnt lgkmcnt(n)` waits until at most n are outstanding; a scalar load in between
    (which may return out of order) leaves only lgkmcnt(0) as proof -- and reports the first instruction that reads the
    destination while the read may still be in flight.  -> [(address of the use, its text, address of the read)]"""
    out = []
    for i, (a, op, args) in enumerate(insts):
        if op not in reads:
            continue
        dst = set(_vregs(args.split(",")[0].strip()))
        later, smem = 0, False      # LDS operations issued after this read; a scalar load seen
        for a2, op2, args2 in insts[i + 1:]:
            if op2.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
                break
            if op2 == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", args2)
                if m and (int(m.group(1)) == 0 or (not smem and int(m.group(1)) <= later)):
                    break                                    # this read has returned
                continue
            ops2 = [o.strip() for o in args2.split(",")] if args2 else []
            used = set()
            for o in _sources(op2, ops2):
                used.update(_vregs(o))
            if dst & used:
                out.append((a2, f"{op2} {args2}", a))
                break
            if op2.startswith("ds_"):
                later += 1
            elif op2.startswith(("s_load", "s_buffer_load")):
                smem = True
            wr = set(_vregs(ops2[0])) if ops2 and not op2.startswith(("ds_write", "global_store", "buffer_store")) else set()
            if dst & wr:
                break                                        # (overwritten: later uses are of the new value)
    return out


def vmem_read_hazards(insts):
    """The same audit for VECTOR-MEMORY loads: a `global_load_*` whose destination is read before an `s_waitcnt vmcnt(n)`
    has brought the counter down far enough.  csrc/mpfft.inc::screen_task issues the sixteen pair-spectrum loads of a
    transform as `asm volatile("global_load_dwordx2 ...")` with a scalar base and waits with a separate
    `asm volatile("s_waitcnt vmcnt(0)")` that carries the sixteen values -- again an order the compiler cannot see.  Loads
    return in order (vmcnt counts loads; stores have their own counter on gfx950's family only where vscnt exists -- here
    stores count in vmcnt too, which can only make a wait stronger).  Walks every load's straight-line successors: a use of
    the destination before a covering wait is a hazard.  -> [(address of the use, its text, address of the load)]"""
    out = []
    for i, (a, op, args) in enumerate(insts):
        if not op.startswith("global_load") or "lds" in op:
            continue
        dst = set(_vregs(args.split(",")[0].strip()))
        later = 0      # vector-memory operations issued after this load
        for a2, op2, args2 in insts[i + 1:]:
            if op2.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
                break
            if op2 == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", args2)
                if m and int(m.group(1)) <= later:
                    break                                    # this load has returned
                continue
            ops2 = [o.strip() for o in args2.split(",")] if args2 else []
            used = set()
            for o in _sources(op2, ops2):
                used.update(_vregs(o))
            if dst & used:
                out.append((a2, f"{op2} {args2}", a))
                break
            if op2.startswith(("global_", "buffer_", "flat_", "scratch_")):
                later += 1
            wr = set(_vregs(ops2[0])) if ops2 and not op2.startswith(("ds_write", "global_store", "buffer_store", "scratch_store")) else set()
            if dst & wr:
                break
    return out


def _sregs(operand):
    m = re.fullmatch(r"s(\d+)", operand)
    if m:
        return [int(m.group(1))]
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", operand)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


VALU_SGPR_WRITERS = ("v_readlane_b32", "v_readfirstlane_b32")


def sgpr_base_hazards(insts, wait_states=5):
    """A vector-memory instruction whose SCALAR base was written by a VALU instruction (v_readlane / v_readfirstlane: SGPR
    spill reloads, hand-made uniform values) fewer than `wait_states` instructions earlier.  gfx9 needs five wait states
    between a VALU write of an SGPR and a VMEM read of it; the compiler's hazard recognizer inserts them for its own
    memory instructions but does not look inside inline asm -- csrc/mpfft.inc's `global_load_dwordx2 v, v_off, s[base]`.
    (Every instruction counts as one wait state here, s_nop n as n + 1: conservative in the right direction only for
    s_nop; a VALU / SALU instruction in between is at least one.)  -> [(address of the load, its text, address of the write)]"""
    out = []
    for i, (a, op, args) in enumerate(insts):
        if not op.startswith(("global_load", "global_store", "global_atomic")) or not args:
            continue
        ops = [o.strip() for o in args.split(",")]
        base = set()
        for o in ops:
            base.update(_sregs(o.split(" ")[0]))
        if not base:
            continue
        states = 0
        for a2, op2, args2 in reversed(insts[max(0, i - 12):i]):
            if op2 in VALU_SGPR_WRITERS and args2:
                if base & set(_sregs(args2.split(",")[0].strip())) and states < wait_states:
                    out.append((a, f"{op} {args}", a2))
                    break
            states += (int(args2.strip(), 0) + 1) if op2 == "s_nop" and args2 else 1
            if states >= wait_states:
                break
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--hazards":
        res = kernel_resources()
        bad = n_reads = 0
        for name in sorted(res):
            insts = kernel_instructions(name + ">:")
            n_reads += sum(op == "ds_read_b64" for _, op, _ in insts)
            h = lds_read_hazards(insts)
            bad += len(h)
            for a, text, pa in h[:5]:
                print(f"{name[:60]}: {a:#x} {text[:70]} uses the destination of the LDS read at {pa:#x}")
            hv = vmem_read_hazards(insts) + sgpr_base_hazards(insts)
            bad += len(hv)
            for a, text, pa in hv[:5]:
                print(f"{name[:60]}: {a:#x} {text[:70]} uses the destination of the global load at {pa:#x}")
        print(f"{len(res)} kernels, {n_reads} ds_read_b64, {bad} hazards")
        sys.exit(1 if bad else 0)
    if len(sys.argv) > 1 and sys.argv[1] == "--pair-loops":
        for (kind, lg), name in sorted(SCREEN_KERNELS.items()):
            print(kind, lg, screen_pair_loop(kind, lg))
        sys.exit(0)
    want = sys.argv[1] if len(sys.argv) > 1 else ""
    for name, r in sorted(kernel_resources().items()):
        if want in name:
            short = subprocess.run(["c++filt", "-p", name], capture_output=True, text=True).stdout.strip() or name
            print(f"{short[:70]:70s} vgpr {r.get('vgpr', -1):4d} spill {r.get('spill', -1):4d} lds {r.get('lds', -1):6d} "
                  f"scratch {r.get('scratch', -1):5d}")
