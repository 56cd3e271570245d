"""Lint of the compiled kernels for a hazard the compiler cannot see (round 4).

Some kernels issue loads from inline assembly (`buffer_load_dwordx4 %0, ...` with the destination as an asm
OUTPUT) and wait for them by hand with a counted `s_waitcnt` tied to the same registers.  Between the two
statements the destination is, for the compiler, an ordinary defined value: if the register allocator copies
it (v_mov, a spill) before the wait, the copy reads a register the load has not written yet and the later
code uses the copy -- silently wrong, and only when the load is late (tests/test_gpu_full_size.py found
exactly that in wino_conv_z_kernel: a 32-register shadow copy of the filter-operand ring around the epilogue).

This script disassembles nothing itself: it reads the `-S` output of hipcc for one source file and reports,
per kernel, every instruction that READS a ring register (a register that is the destination of an
asm-issued buffer_load_dwordx4) other than the MFMAs that consume it.

    python tools/isa_lint.py <file.s> [kernel-name-substring]      -> exit code 1 when something is found
"""
import re
import sys


def expand(tok):
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def kernels(text):
    cur, name = [], None
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            if name:
                yield name, cur
            name, cur = m.group(1), []
        elif name is not None:
            cur.append(line)
            if line.startswith("\t.set " + name + ".uses_flat_scratch") or line.startswith(".Lfunc_end"):
                yield name, cur
                name, cur = None, []
    if name:
        yield name, cur


def lint(lines):
    ring, in_asm, first = set(), False, None
    for no, l in enumerate(lines):
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        elif in_asm:
            m = re.match(r"\s*(?:buffer_load_dwordx4|ds_read_b64_tr_b16)\s+(v\[\d+:\d+\]),", l)
            if m and " lds" not in l:
                ring |= expand(m.group(1))
                first = no if first is None else first
    bad = []
    if not ring:
        return ring, bad
    # linear scan: a ring register is "asm-owned" from an asm load that writes it until an ordinary instruction
    # overwrites it (the allocator may use a slot as a temporary between its consumption and its refill)
    owned, owned_ds, in_asm = set(), set(), False
    for no, l in enumerate(lines):
        if "#ASMSTART" in l:
            in_asm = True
            continue
        if "#ASMEND" in l:
            in_asm = False
            continue
        code = l.split(";")[0].strip()
        if not code or code.startswith(".") or code.endswith(":"):
            continue
        parts = code.split(None, 1)
        op = parts[0]
        if in_asm:
            m = re.match(r"buffer_load_dwordx4\s+(v\[\d+:\d+\]),", code)
            if m and " lds" not in code:
                owned |= expand(m.group(1))
            m = re.match(r"ds_read_b64_tr_b16\s+(v\[\d+:\d+\]),", code)
            if m:
                owned_ds |= expand(m.group(1))
            if code.startswith("s_waitcnt") and re.search(r"\blgkmcnt\(0\)", code):
                owned_ds.clear()          # the hand-written wait: LDS reads have returned
            if code.startswith("s_waitcnt") and re.search(r"\bvmcnt\(0\)", code):
                owned.clear()             # exactly vmcnt(0): a wait on any other count frees nothing here
            continue
        if len(parts) < 2:
            continue
        ops = [t.strip() for t in parts[1].split(",")]
        stores = op.startswith(("scratch_store", "buffer_store", "global_store", "ds_write", "ds_store"))
        srcs = ops if stores else ops[1:]
        read = set()
        for t in srcs:
            read |= expand(t.split()[0]) if t else set()
        if (not op.startswith("v_mfma") and (read & owned)) or (read & owned_ds):
            bad.append((no, code))
        if not stores and not op.startswith(("s_", "v_cmp", "v_readfirstlane", "v_readlane")):
            dst = expand(ops[0].split()[0]) if ops and ops[0] else set()
            owned -= dst
            owned_ds -= dst
    return ring, bad


def sgprs(tok):
    tok = tok.strip()
    m = re.match(r"s\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"s(\d+)$", tok)
    if m:
        return {int(m.group(1))}
    return {"vcc"} if tok in ("vcc", "vcc_lo", "vcc_hi") else set()


def lint_scalar_operands(lines, need=5):
    """Round 6.  A scalar register written by a VALU instruction (v_readlane restoring a spilled SGPR, v_readfirstlane,
    v_cmp into an SGPR pair) may not be read by a vector-memory instruction for 5 wait states.  hipcc inserts the
    s_nop for its own loads; it cannot for a buffer_load / buffer_store inside an inline-asm statement, whose operands
    it only sees as "s" constraints -- conv3x3_split.hip's first version read the PREVIOUS value of a spilled offset in
    the first load behind every restore (wrong filter rows for one wave's half of an item, only when the register
    allocator spilled, i.e. only in the persistent form of the kernel).  Reports (line, instruction, register, wait
    states seen) for every asm VMEM instruction closer than `need` wait states to such a write."""
    recent = []          # [wait states since, registers]
    bad, in_asm = [], False
    for no, l in enumerate(lines):
        if "#ASMSTART" in l:
            in_asm = True
            continue
        if "#ASMEND" in l:
            in_asm = False
            continue
        code = l.split(";")[0].strip()
        if not code or code.startswith(".") or code.endswith(":"):
            continue
        parts = code.split(None, 1)
        op = parts[0]
        ops = [t.strip() for t in parts[1].split(",")] if len(parts) > 1 else []
        if in_asm and op.startswith(("buffer_load", "buffer_store", "global_load", "global_store")):
            read = set()
            for t in ops:
                for w in t.split():
                    read |= sgprs(w)
            for ws, regs in recent:
                hit = read & regs
                if hit and ws < need:
                    bad.append((no, code, sorted(hit, key=str)[0], ws))
        ws = int(ops[0]) + 1 if op == "s_nop" and ops else 1
        recent = [[w + ws, r] for w, r in recent if w + ws < 16]
        written = set()
        if op.startswith(("v_readlane_b32", "v_readfirstlane_b32")) and ops:
            written = sgprs(ops[0])
        elif op.startswith("v_cmp") and ops and op.endswith("_e64"):
            written = sgprs(ops[0])
        elif op.startswith("v_cmp"):
            written = {"vcc"}
        if written:
            recent.append([0, written])
    return bad


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    rc = 0
    for name, lines in kernels(text):
        if want not in name:
            continue
        ring, bad = lint(lines)
        hz = lint_scalar_operands(lines)
        if not ring and not hz:
            continue
        print("%s: %d ring registers, %d non-MFMA reads, %d asm VMEM reads of a VALU-written SGPR inside 5 wait states"
              % (name, len(ring), len(bad), len(hz)))
        for no, code in bad[:20]:
            print("    +%d  %s" % (no, code))
        for no, code, reg, ws in hz[:20]:
            print("    +%d  %s   <- s%s written %d wait state(s) earlier" % (no, code, reg, ws))
        rc |= 1 if (bad or hz) else 0
    return rc


if __name__ == "__main__":
    sys.exit(main())
