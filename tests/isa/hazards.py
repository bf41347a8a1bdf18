"""Wait-state checker for hand-written gfx950 blocks (TEST INFRASTRUCTURE).

hipcc pads the data hazards of the instructions it schedules with `s_nop`, but not those inside an `asm`
statement.  This walks straight-line instruction sequences of `hipcc -S` output (every path through branch
targets is followed up to the hazard horizon) and reports producer -> consumer pairs that are closer than the
rule allows.  Rules (wait states = issue slots between the two instructions, `s_nop N` counts N + 1):

  R1  VALU writes an SGPR pair / VCC  ->  VALU reads it (mask, carry-in or scalar source)            2
  R2  VALU writes an SGPR / VCC       ->  v_readlane / v_writelane lane select                        4
  R3  VALU writes an SGPR             ->  vector memory instruction reads it (saddr / descriptor)     5
  R5  VALU writes a VGPR              ->  DPP reads it                                                2
      VALU writes EXEC                ->  DPP                                                         5
  R7  VALU writes a VGPR              ->  v_readlane / v_readfirstlane reads it                       1
  R9  soft clauses (XNACK replay): in a run of consecutive vector loads no load may write a VGPR that a load
      of the run uses as address (hipcc breaks such runs with `s_nop 0`)

`calibrate()` checks the rule set against the compiler's own code: every `s_nop` hipcc placed outside the asm
blocks must be explained by a rule (otherwise a rule is missing) and compiler code must not violate a rule
(otherwise a rule is too strict).
"""
import re
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gfx950_emu as emu  # noqa: E402

HORIZON = 6


def _regs(o):
    k = o[0]
    if k in ("s", "v", "a"):
        return [(k, o[1] + j) for j in range(o[2])]
    if k == "exec":
        return [("exec", 0)]
    if k in ("exec_lo", "exec_hi"):
        return [("exec", 0)]
    return []


def _is_valu(i):
    return i.op.startswith("v_")


def _is_vmem(i):
    return i.op.startswith(("global_", "flat_", "buffer_", "scratch_"))


def _defs_uses(i):
    """-> (sgpr/vcc/exec defs by VALU, vgpr defs by VALU, uses as list of (reg, role))"""
    op = i.op
    ops = i.ops
    sdefs, vdefs, uses = [], [], []
    if _is_valu(i):
        base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
        if base.startswith("v_cmp"):
            sdefs += _regs(ops[0])
            srcs = ops[1:]
        elif base in ("v_readlane_b32", "v_readfirstlane_b32"):
            sdefs += _regs(ops[0])
            srcs = ops[1:]
            for r in _regs(ops[1]):
                uses.append((r, "lane_src"))
            if len(ops) > 2:
                for r in _regs(ops[2]):
                    uses.append((r, "lane_sel"))
            return sdefs, vdefs, uses
        elif base == "v_writelane_b32":
            vdefs += _regs(ops[0])
            for r in _regs(ops[1]):
                uses.append((r, "valu_src"))
            for r in _regs(ops[2]):
                uses.append((r, "lane_sel"))
            return sdefs, vdefs, uses
        elif base in ("v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_addc_co_u32", "v_subb_co_u32",
                      "v_subbrev_co_u32", "v_mad_u64_u32", "v_mad_i64_i32"):
            vdefs += _regs(ops[0])
            sdefs += _regs(ops[1])
            srcs = ops[2:]
        else:
            vdefs += _regs(ops[0]) if ops else []
            srcs = ops[1:]
        for o in srcs:
            for r in _regs(o):
                uses.append((r, "valu_src"))
        if op.endswith("_dpp"):
            for r in _regs(ops[1]):
                uses.append((r, "dpp_src"))
            uses.append((("exec", 0), "dpp_exec"))
    elif _is_vmem(i):
        for o in ops:
            if o[0] == "s":
                for r in _regs(o):
                    uses.append((r, "vmem_sgpr"))
    return sdefs, vdefs, uses


def _wait(i):
    if i.op == "s_nop":
        return (i.ops[0][1] if i.ops else 0) + 1
    return 1


def check(insts, labels, lo=0, hi=None, report_nops=False):
    """Checks insts[lo:hi].  Returns (violations, unexplained_nops)."""
    hi = len(insts) if hi is None else hi
    need = {"valu_src": 2, "lane_sel": 4, "vmem_sgpr": 5}
    viol = []
    explained = set()

    def successors(k):
        i = insts[k]
        out = []
        if i.target is not None:
            if i.target in labels:
                out.append(labels[i.target])
            if i.op != "s_branch":
                out.append(k + 1)
        elif i.op != "s_endpgm":
            out.append(k + 1)
        return [x for x in out if lo <= x < hi]

    for k in range(lo, hi):
        i = insts[k]
        sdefs, vdefs, _ = _defs_uses(i)
        if not sdefs and not vdefs:
            continue
        # walk forward along every path up to the horizon
        stack = [(s, 0, (k,)) for s in successors(k)]
        seen = set()
        while stack:
            j, dist, path = stack.pop()
            if dist >= HORIZON or (j, dist) in seen:
                continue
            seen.add((j, dist))
            c = insts[j]
            _, _, uses = _defs_uses(c)
            for reg, role in uses:
                req = None
                if reg in sdefs and role in need:
                    req = need[role]
                elif reg in sdefs and reg == ("exec", 0) and role == "dpp_exec":
                    req = 5
                elif reg in vdefs and role == "dpp_src":
                    req = 2
                elif reg in vdefs and role == "lane_src":
                    req = 1
                if req is not None:
                    if dist < req:
                        viol.append((i.line, i.text, c.line, c.text, role, dist, req))
                    else:
                        for q in path[1:]:
                            if insts[q].op == "s_nop" and dist - _wait(insts[q]) < req:
                                explained.add(q)
            # a redefinition by a non-VALU instruction ends the hazard for that register; keep it simple: go on
            nd = dist + _wait(c)
            for s in successors(j):
                stack.append((s, nd, path + (j,)))
    # R9: runs of consecutive vector loads
    k = lo
    while k < hi:
        if not (_is_vmem(insts[k]) and "_load_" in insts[k].op):
            k += 1
            continue
        e = k
        addr, dest = set(), set()
        while e < hi and _is_vmem(insts[e]) and "_load_" in insts[e].op:
            c = insts[e]
            d = set(_regs(c.ops[0]))
            a = set(r for r in _regs(c.ops[1]) if r[0] == "v")
            if (d & (addr | a) and e > k) or (dest & a):
                viol.append((insts[k].line, insts[k].text, c.line, c.text, "soft_clause", 0, 1))
            addr |= a
            dest |= d
            e += 1
        if e < hi and insts[e].op == "s_nop" and e + 1 < hi and _is_vmem(insts[e + 1]) and "_load_" in insts[e + 1].op:
            explained.add(e)
        k = e + 1
    nops = [q for q in range(lo, hi) if insts[q].op == "s_nop" and q not in explained]
    return viol, nops


def asm_regions(text):
    """line ranges (1-based, inclusive) between ;;#ASMSTART and ;;#ASMEND"""
    out, start = [], None
    for ln, raw in enumerate(text.splitlines(), 1):
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            start = ln
        elif t.startswith(";;#ASMEND") and start is not None:
            out.append((start, ln))
            start = None
    return out


def check_kernel(text, entry):
    prog = emu.Program(text, entry)
    regions = asm_regions(text)
    in_asm = lambda line: any(a <= line <= b for a, b in regions)  # noqa: E731
    viol, nops = check(prog.insts, prog.labels)
    asm_viol = [v for v in viol if in_asm(v[0]) or in_asm(v[2])]
    cc_viol = [v for v in viol if not (in_asm(v[0]) or in_asm(v[2]))]
    cc_nops = [prog.insts[q] for q in nops if not in_asm(prog.insts[q].line)]
    return asm_viol, cc_viol, cc_nops, prog


if __name__ == "__main__":
    import lz4_kernel as lk

    for src, needle in (("lz4_compress.hip", "lz4_compress_l2_kernelILb1E"), ("snappy_compress.hip", "snappy_compress_kernelILb1E")):
        text = lk.compile_asm(src)
        entry = lk.find_kernel(text, needle)
        asm_viol, cc_viol, cc_nops, prog = check_kernel(text, entry)
        print("==", src)
        print("violations inside asm blocks:", len(asm_viol))
        for v in asm_viol:
            print("  line %d: %s  ->  line %d: %s   (%s: %d < %d)" % v)
        print("violations in compiler code (rule too strict?):", len(cc_viol))
        for v in cc_viol[:20]:
            print("  line %d: %s  ->  line %d: %s   (%s: %d < %d)" % v)
        print("compiler s_nops no rule explains (rule missing?):", len(cc_nops))
        for i in cc_nops[:40]:
            k = prog.insts.index(i)
            ctx = " | ".join(x.text for x in prog.insts[max(0, k - 2):k + 3])
            print("  line %d: %s" % (i.line, ctx))
