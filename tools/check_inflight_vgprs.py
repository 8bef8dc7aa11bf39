"""Offline check for LDS reads whose wait is issued later by hand (csrc/attention.hip TrFrag / lds_read_f4, csrc/gemm_wgrad.hip
SAVIT_TR_READ, the park / slab reads of csrc/gemm_tn.hip): an inline-asm `ds_read_b64_tr_b16` / `ds_read_b128` names its
destination VGPRs as an OUTPUT, so hipcc believes they are defined the moment the asm statement is issued and may read, copy
(`v_mov`, a live-range split) or re-use them before the `s_waitcnt lgkmcnt(N)` that really retires the read - the copy then holds
stale data.  Nothing in the language prevents it; these kernels sit at 250+ VGPRs where the allocator is under the most pressure.

The check, on the disassembly of the built object: model the LDS queue in issue order (LDS operations return in order, so
`s_waitcnt lgkmcnt(N)` retires all but the N youngest of them; scalar-memory loads share the counter and can only make a wait
longer), and flag every instruction that READS or WRITES a VGPR which is the destination of an LDS read still in the queue.  It covers compiler-generated reads too (which hipcc
always waits for correctly), so a clean result is a statement about the whole kernel.

Control flow is followed (hipcc lays loop bodies out of line): basic blocks from the branch targets, a forward data-flow pass to a
fixed point (a register is in flight at a join if it is on any incoming path), so a read issued at the bottom of an iteration and
consumed - or clobbered - at the top of the next, or behind the loop exit, is seen.

usage: check_inflight_vgprs.py kernel.s [kernel-name-regex]     (ISA text: `llvm-objdump -d` of the gfx950 code object)"""
import re
import sys

_VR = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
_LGKM = re.compile(r"lgkmcnt\((\d+)\)")
_ADDR = re.compile(r"//\s*([0-9A-Fa-f]+):")
_TARGET = re.compile(r"<[^>+]+\+0x([0-9A-Fa-f]+)>")


def _vregs(text):
    out = set()
    for m in _VR.finditer(text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def _parse(isa_text):
    """-> [(op, operand text, address or None, branch-target offset or None, raw line)] for the instruction lines"""
    ins = []
    for raw in isa_text.splitlines():
        code = raw.split("//")[0].strip()
        if not code or code.startswith((";", ".")) or code.endswith(":") or re.match(r"^[0-9a-f]+ <", code):
            continue
        parts = code.split(None, 1)
        op, rest = parts[0], (parts[1] if len(parts) > 1 else "")
        if not re.match(r"^[a-z_0-9]+$", op):
            continue
        am = _ADDR.search(raw)
        tm = _TARGET.search(rest) if op.startswith(("s_cbranch", "s_branch")) else None
        ins.append((op, rest, int(am.group(1), 16) if am else None, int(tm.group(1), 16) if tm else None, code))
    return ins


def _is_lds_read(op):
    return op.startswith("ds_read") or op.startswith("ds_load") or op in ("ds_bpermute_b32", "ds_permute_b32", "ds_swizzle_b32") or \
        (op.startswith("ds_") and "_rtn" in op)


MAXD = 16  # lgkmcnt is a 4-bit counter: a read with 16 or more LGKM operations issued behind it can no longer be told apart


def _transfer(depth, op, rest, code, bad, where):
    """One instruction over the abstract LDS state.  depth[v] = the LEAST number of LDS operations issued behind the in-flight LDS
    read that writes VGPR v, over all paths reaching this point.  LDS operations return in order among themselves, and lgkmcnt counts
    them together with scalar-memory loads, so `s_waitcnt lgkmcnt(N)` (at most N operations of both kinds outstanding, hence at most
    N LDS operations) retires every read with depth >= N whatever scalar loads are in flight beside them (those can only make the
    wait longer; their own out-of-order return is tools/check_inflight_sgprs.py's subject).  Returns the new state."""
    if op == "s_waitcnt":
        m = _LGKM.search(rest)
        if m:
            n = int(m.group(1))
            depth = {v: k for v, k in depth.items() if k < n}
        return depth
    if op.startswith("ds_"):
        ops = [o.strip() for o in rest.split(",")]
        if _is_lds_read(op):
            dest = _vregs(ops[0]) if ops else set()
            if bad is not None and _vregs(",".join(ops[1:])) & depth.keys():
                bad.append((where, code))
            # (a destination that overlaps an in-flight one is a write-after-write inside the in-order LDS queue: legal)
        else:
            dest = set()
            if bad is not None and _vregs(rest) & depth.keys():
                bad.append((where, code))
        depth = {v: min(MAXD, k + 1) for v, k in depth.items()}
        for v in dest:
            depth[v] = 0
        return depth
    if depth and bad is not None and _vregs(rest) & depth.keys():
        bad.append((where, code))
    return depth


def _merge(a, b):
    """join of two states: a VGPR is in flight if it is on either path, at the smaller depth"""
    if a is None:
        return dict(b), True
    depth, changed = dict(a), False
    for v, k in b.items():
        if v not in depth or k < depth[v]:
            depth[v] = k
            changed = True
    return depth, changed


def violations(isa_text):
    """-> (LDS reads seen, [(instruction index, offending instruction)]) for ONE kernel's disassembly.  Control flow is followed:
    basic blocks from the branch targets (s_branch / s_cbranch_*: target = address + 4 + 4 * simm16), a forward data-flow pass to a
    fixed point with the join above, then one reporting pass over every block with its converged entry state."""
    ins = _parse(isa_text)
    n = len(ins)
    if n == 0:
        return 0, []
    at = {a: i for i, (_, _, a, _, _) in enumerate(ins) if a is not None}
    succ = [[] for _ in range(n)]
    for i, (op, rest, addr, _, _) in enumerate(ins):
        is_br = op.startswith(("s_cbranch_", "s_branch"))
        tgt = None
        if is_br and addr is not None:
            m = re.match(r"\s*(-?\d+)", rest)
            if m:
                imm = int(m.group(1)) & 0xFFFF
                imm -= 0x10000 if imm >= 0x8000 else 0
                tgt = at.get(addr + 4 + 4 * imm)
        if tgt is not None:
            succ[i].append(tgt)
        if op in ("s_endpgm", "s_branch") or op.startswith(("s_setpc", "s_swappc")):
            continue
        if i + 1 < n:
            succ[i].append(i + 1)
    state_in = [None] * n
    state_in[0] = {}
    work = [0]
    while work:
        i = work.pop()
        out = _transfer(state_in[i], ins[i][0], ins[i][1], ins[i][4], None, i)
        for j in succ[i]:
            merged, changed = _merge(state_in[j], out)
            if changed or state_in[j] is None:
                state_in[j] = merged
                work.append(j)
    bad, reads = [], 0
    for i, (op, rest, _, _, code) in enumerate(ins):
        if op.startswith("ds_") and _is_lds_read(op):
            reads += 1
        if state_in[i] is not None:
            _transfer(state_in[i], op, rest, code, bad, i)
    last = max((i for i in range(n) if ins[i][0] == "s_endpgm"), default=n - 1)  # what follows the last s_endpgm is padding
    violations.last_unreached = sum(1 for i in range(last + 1) if state_in[i] is None and ins[i][0] not in ("s_nop", "s_code_end"))
    return reads, bad


violations.last_unreached = 0  # instructions of the last kernel scanned that no path from its entry reaches (padding aside: expect 0)


def split_kernels(isa_text):
    """{symbol: its disassembly} of an `llvm-objdump -d` listing"""
    out = {}
    for blk in re.split(r"\n(?=[0-9a-f]+ <)", isa_text):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", blk.strip())
        if m:
            out[m.group(1)] = blk
    return out


if __name__ == "__main__":
    text = open(sys.argv[1]).read()
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    rc = 0
    for name, body in split_kernels(text).items():
        if pat is not None and not pat.search(name):
            continue
        n, bad = violations(body)
        if bad:
            rc = 1
        print(f"{name}: {n} LDS reads, {len(bad)} in-flight VGPR violations")
        for w, c in bad[:8]:
            print("   touches an in-flight LDS destination:", c)
    sys.exit(rc)
