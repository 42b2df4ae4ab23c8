#!/usr/bin/env python
"""Build-time check of the hand-scheduled kernels: a `global_load_dwordx4` written as inline asm is invisible
to the compiler's wait-count pass -- the kernel waits for it with its own `s_waitcnt vmcnt(n)`.  The compiler therefore believes
the destination registers are defined the moment the asm statement is passed.  This tool walks the ISA of one kernel (hipcc -S
output) linearly, keeps the queue of issued vector-memory loads, retires them at every s_waitcnt vmcnt(n) (loads return in order),
and reports every instruction that READS or WRITES a register while a load into it is still in flight.  Loop bodies are walked
twice (state carried over the back edge).  Forward branches are ignored (both sides are walked in text order: conservative).

    python vamb_amd/csrc/isa_pending_loads.py vae.s [<mangled-name-substring>] [--all-loads]

Only loads between the ;;#ASMSTART / ;;#ASMEND markers of inline asm are tracked with their registers (the compiler waits for its
own loads before it touches their registers, by construction); every vector-memory operation takes its place in the in-order
queue that s_waitcnt vmcnt(n) retires.  --all-loads tracks the compiler's loads as well (noisy: the linear walk does not follow
its branches).  Because the queue is retired by COUNT, the check covers the hand-counted waits themselves: a wait that waits for
too little leaves the consuming instruction (an MFMA reading a tile register) flagged.  Limitation: control flow is not analysed --
a wait inside a conditionally executed region is assumed to have run; the kernels checked here wait unconditionally in front of
every consumer.
"""
import re
import sys


def regs(tok):
    """vector registers named by an operand token: v12, v[4:7], a3, a[0:3]"""
    out = set()
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]", tok):
        for i in range(int(m.group(2)), int(m.group(3)) + 1):
            out.add(m.group(1) + str(i))
    for m in re.finditer(r"\b([va])(\d+)\b", tok):
        out.add(m.group(1) + m.group(2))
    return out


def parse(line):
    s = line.split(";")[0].strip()
    if not s or s.startswith(".") or s.endswith(":"):
        return None
    parts = s.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return op, ops


def dst_src(op, ops):
    """(written, read) vector registers of an instruction (approximation: first operand is the destination, except stores)"""
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "ds_write", "ds_store")):
        return set(), set().union(*[regs(o) for o in ops]) if ops else set()
    if op.startswith(("s_", "v_cmp", "v_cmpx")):
        return set(), set().union(*[regs(o) for o in ops]) if ops else set()
    if not ops:
        return set(), set()
    w = regs(ops[0])
    r = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
    if op.startswith("v_mfma") or op.startswith(("v_fmac", "v_mac", "v_pk_fma")) or "accvgpr" in op:
        pass
    if op.startswith(("v_fmac", "v_mac", "v_dot")):
        r |= w
    return w, r


def kernels(lines):
    """(name, body lines) of every kernel of a hipcc -S dump"""
    i = 0
    while i < len(lines):
        if re.match(r"^_Z\w*:", lines[i]):
            end = i
            while end < len(lines) and "s_endpgm" not in lines[end] and not lines[end].startswith("\t.section"):
                end += 1
            yield lines[i].split(":")[0], lines[i:end + 1]
            i = end
        i += 1


def check_kernel(body, all_loads=False, out=print):
    """number of hazards in one kernel body; every hazard is reported through `out`"""
    labels = {l.split(":")[0].strip(): i for i, l in enumerate(body) if re.match(r"^\.LBB\w+:", l.strip())}
    # back edges: a branch to a label above it
    back = []
    for i, l in enumerate(body):
        p = parse(l)
        if p and p[0].startswith("s_cbranch") or (p and p[0] == "s_branch"):
            tgt = p[1][-1]
            if tgt in labels and labels[tgt] < i:
                back.append((labels[tgt], i))
    order = []
    i = 0
    done_loops = set()
    while i < len(body):
        order.append(i)
        hit = [b for b in back if b[1] == i and b not in done_loops]
        if hit:
            done_loops.add(hit[0])
            i = hit[0][0]      # walk the loop body a second time
            continue
        i += 1
    pending = []   # queue of (set of dst regs, line index) in issue order
    problems = 0
    # only loads written as inline asm are tracked with their registers (the compiler waits for its own loads before it touches
    # their registers, by construction); the compiler's loads still take their place in the in-order queue (empty register set)
    in_asm = set()
    inside = False
    for i, l in enumerate(body):
        if "#ASMSTART" in l:
            inside = True
        elif "#ASMEND" in l:
            inside = False
        elif inside:
            in_asm.add(i)
    for idx in order:
        p = parse(body[idx])
        if not p:
            continue
        op, ops = p
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", body[idx])
            if m:
                n = int(m.group(1))
                while len(pending) > n:
                    pending.pop(0)
            continue
        if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
            lds_dma = "_lds_" in op or " lds" in body[idx]      # LDS-DMA: no register destination, every operand is an address
            w = set() if lds_dma else regs(ops[0])
            r = set().union(*[regs(o) for o in (ops if lds_dma else ops[1:])])
            for dst, li in pending:
                if r & dst:
                    out(f"line {idx}: `{body[idx].strip()[:90]}` uses as ADDRESS {sorted(r & dst)} that the load at line {li} is still writing")
                    problems += 1
                if w & dst:
                    out(f"line {idx}: `{body[idx].strip()[:90]}` targets {sorted(w & dst)} while the load at line {li} into them is still in flight")
                    problems += 1
            pending.append((w if (idx in in_asm or all_loads) else set(), idx))
            continue
        if op.startswith(("global_store", "buffer_store", "flat_store", "global_atomic")):
            pending.append((set(), idx))   # stores count in vmcnt on gfx9 as well
        w, r = dst_src(op, ops)
        for dst, li in pending:
            if (w | r) & dst:
                kind = "WRITES" if w & dst else "READS"
                out(f"line {idx}: `{body[idx].strip()[:90]}` {kind} {sorted((w | r) & dst)} while the load at line {li} (`{body[li].strip()[:60]}`) may still be in flight")
                problems += 1
    return problems


def check_file(path, pattern=None, all_loads=False, out=print):
    """(kernels checked, kernels with asm loads, hazards) of a hipcc -S dump; pattern: substring of the mangled name, or None"""
    lines = open(path).read().split("\n")
    n_k = n_asm = total = 0
    for name, body in kernels(lines):
        if pattern and pattern not in name + ":":
            continue
        n_k += 1
        has_asm = False
        inside = False
        for l in body:
            if "#ASMSTART" in l:
                inside = True
            elif "#ASMEND" in l:
                inside = False
            elif inside and re.match(r"\s*(global|buffer|flat)_load", l) and "lds" not in l:
                has_asm = True
                break
        if not (has_asm or all_loads):
            continue
        n_asm += 1
        p = check_kernel(body, all_loads, out)
        if p:
            out(f"{p} hazard(s) in {name}")
        total += p
    return n_k, n_asm, total


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = args[0]
    pattern = args[1] if len(args) > 1 else None
    n_k, n_asm, total = check_file(path, pattern, "--all-loads" in sys.argv)
    print(f"{path}: {n_k} kernels, {n_asm} with hand-waited (inline asm) register loads, {total} hazard(s)")
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
