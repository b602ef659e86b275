#!/usr/bin/env python3
"""gcn_sched.py — a post-register-allocation instruction scheduler for straight-line pieces of gfx950 (CDNA4) assembly.

Why: LLVM's scheduling model for gfx940/gfx950 prices an FP64 VALU instruction at its ISSUE cost (one quad cycle), so it
sees no reason to interleave independent chains; the hardware issues a wave's instructions in order and a dependent
FP64 instruction waits ≈ 8+ cycles for its operand (scratch/ubench_lat.hip).  A wave that has its SIMD to itself (a
latitude slab of a strongly scaled run) therefore runs the Monin–Obukhov iteration at ≈ 1750 cycles although it issues
in ≈ 580.  This tool re-orders the instructions of the hot loop's basic blocks of the compiler's own output — same
instructions, same registers, same results bit for bit — with the measured latencies, recomputes the s_waitcnt
lgkmcnt() counts for the new order of the LDS reads and re-inserts the hazard no-ops of the gfx940 family.

Scope (deliberately narrow): blocks without stores / global memory / LDS writes / exec writes / scalar memory;
anything it does not understand makes it leave the block alone.

  gcn_sched.py in.s out.s --function <substring> [--blocks-with ds_read_b128] [--report]
"""
import re
import sys
import argparse
from collections import defaultdict

# ---------------------------------------------------------------------------------------------
# machine model (cycles; scratch/ubench_lat.hip, scratch/ubench_valu.hip on MI355X)
# ---------------------------------------------------------------------------------------------
# issue: cycles the wave's issue port is busy; latency: issue → a dependent instruction may issue
# One wave alone on its SIMD (scratch/ubench_lat.hip, in s_memtime ticks): it issues at most one instruction per ≈ 4
# ticks whatever the type; a dependent VALU instruction of ANY type follows its producer after ≈ 9; v_rcp/v_rsq_f64 ≈ 20.5;
# f32 transcendentals ≈ 13 (+ the mandatory wait state); an LDS read's data ≈ 64 after its issue.
MODEL = dict(
    f64=(4, 9), f64_trans=(16, 21), f32=(4, 9), f32_trans=(8, 13), int32=(4, 9), int_vop3=(4, 9), cvt=(4, 9),
    mov64=(4, 9), cmp64=(4, 9), cmp32=(4, 9), salu=(4, 6), lds=(4, 64), nop=(4, 4), other=(4, 9))


class Inst:
    __slots__ = ("text", "op", "defs", "uses", "kind", "is_lds", "is_valu", "is_trans", "index", "comment_only", "raw_lines")

    def __init__(self, text):
        self.text = text
        self.raw_lines = [text]


REG_RANGE = re.compile(r"\b([vsa])\[(\d+):(\d+)\]")
REG_ONE = re.compile(r"\b([vs])(\d+)\b")


def regs_of(tok):
    """registers named in one operand token → set of 'v12', 's3', 'vcc_lo', 'vcc_hi', 'exec_lo', …"""
    out = set()
    t = tok
    for m in REG_RANGE.finditer(t):
        for n in range(int(m.group(2)), int(m.group(3)) + 1):
            out.add(f"{m.group(1)}{n}")
    t2 = REG_RANGE.sub(" ", t)
    for m in REG_ONE.finditer(t2):
        out.add(f"{m.group(1)}{m.group(2)}")
    if re.search(r"\bvcc\b", t2):
        out |= {"vcc_lo", "vcc_hi"}
    for h in ("vcc_lo", "vcc_hi", "exec_lo", "exec_hi", "m0", "scc"):
        if re.search(rf"\b{h}\b", t2):
            out.add(h)
    if re.search(r"\bexec\b", t2):
        out |= {"exec_lo", "exec_hi"}
    return out


def split_operands(s):
    """top-level comma split (brackets protect v[1:2])"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "[(":
            depth += 1
        elif ch in "])":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


FMAC = re.compile(r"^v_(fmac|mac|pk_fmac)_")
TRANS = re.compile(r"^v_(rcp|rsq|sqrt|log|exp|sin|cos)_")


def classify(op):
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "lds"
    if op in ("s_nop",):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    if TRANS.match(op):
        return "f64_trans" if "f64" in op else "f32_trans"
    if op.startswith("v_cmp"):
        return "cmp64" if ("f64" in op or "64" in op.split("_")[-2:][0]) else "cmp32"
    if op.startswith("v_cvt"):
        return "cvt"
    if op.startswith("v_mov_b64") or op.startswith("v_lshl_add_u64") or op.startswith("v_lshlrev_b64"):
        return "mov64"
    if "f64" in op:
        return "f64"
    if op.startswith(("v_lshl_add_u32", "v_add3_u32", "v_bfe", "v_and_or", "v_mul_lo", "v_mul_hi", "v_mad_", "v_lshl_or", "v_cndmask_b32_e64")):
        return "int_vop3"
    if re.search(r"_f32|_f16", op):
        return "f32"
    if op.startswith("v_"):
        return "int32"
    return "other"


UNSAFE = re.compile(r"^(ds_write|ds_store|ds_add|ds_|global_|flat_|buffer_|scratch_|s_load|s_store|s_buffer|s_barrier|s_setprio|s_sleep|s_sendmsg|s_endpgm|"
                    r"s_cbranch|s_branch|s_setpc|s_swappc|s_getpc|v_readfirstlane|v_writelane|v_permlane|ds_bpermute|ds_permute|s_memtime|s_getreg|s_setreg|"
                    r"v_mfma|v_smfma|s_set_gpr_idx|v_movrel|v_div_scale|v_div_fmas|v_mbcnt)")


def parse_inst(line):
    """one assembly line → Inst (defs / uses as register-name sets), or None for something that pins the block"""
    code = line.split(";")[0].strip()
    ins = Inst(line)
    if not code:
        ins.comment_only = True
        ins.op = ""
        ins.defs, ins.uses, ins.kind = set(), set(), "nop"
        ins.is_lds = ins.is_valu = ins.is_trans = False
        return ins
    ins.comment_only = False
    m = re.match(r"(\S+)\s*(.*)", code)
    op, rest = m.group(1), m.group(2)
    ins.op = op
    if UNSAFE.match(op) and not op.startswith("ds_read"):
        return None
    if "_dpp" in op or "_sdwa" in op or re.search(r"\b(row_|quad_perm|wave_|bank_mask|dst_sel|src0_sel)", rest):
        return None   # DPP / SDWA carry their own hazards (VALU → DPP read: 2 wait states): such pieces are left alone
    ops = split_operands(rest)
    ins.kind = classify(op)
    ins.is_lds = ins.kind == "lds"
    ins.is_valu = op.startswith("v_")
    ins.is_trans = bool(TRANS.match(op))
    defs, uses = set(), set()
    if op == "s_waitcnt" or op == "s_nop":
        ins.defs, ins.uses = set(), set()
        return ins
    if op.startswith("ds_read"):
        # ds_read_b128 vdst, vaddr [offset:…]; ds_read2… vdst, vaddr offset0:… offset1:…
        toks = split_operands(rest)
        defs |= regs_of(toks[0])
        addr = toks[1].split()[0]
        uses |= regs_of(addr)
        uses |= {"exec_lo", "exec_hi"}
        ins.defs, ins.uses = defs, uses
        return ins
    if not ops:
        return None
    if op.startswith("s_"):
        if op.startswith("s_cmp") or op.startswith("s_bitcmp"):
            defs.add("scc")
            for o in ops:
                uses |= regs_of(o)
        else:
            defs |= regs_of(ops[0])
            for o in ops[1:]:
                uses |= regs_of(o)
            if re.match(r"s_(and|or|xor|andn2|orn2|nand|nor|xnor|add|sub|addc|subb|lshl|lshr|ashr|min|max|mul|bfe|not|abs|bcnt|ff|flbit|absdiff|lshl\d_add)_", op) or "saveexec" in op:
                defs.add("scc")
            if op.startswith(("s_addc", "s_subb", "s_cselect", "s_cmov")):
                uses.add("scc")
            if "saveexec" in op:
                defs |= {"exec_lo", "exec_hi"}
                uses |= {"exec_lo", "exec_hi"}
        ins.defs, ins.uses = defs, uses
        return ins
    if not op.startswith("v_"):
        return None
    # VALU
    uses |= {"exec_lo", "exec_hi"}
    if op.startswith("v_cmpx"):
        return None
    if op.startswith("v_cmp"):
        defs |= regs_of(ops[0])          # vcc or an SGPR pair, spelled out in both encodings
        for o in ops[1:]:
            uses |= regs_of(o)
    elif op.startswith("v_readlane"):
        defs |= regs_of(ops[0])
        for o in ops[1:]:
            uses |= regs_of(o)
    else:
        defs |= regs_of(ops[0])
        for o in ops[1:]:
            uses |= regs_of(o)
        if FMAC.match(op):
            uses |= regs_of(ops[0])
        if op.startswith("v_cndmask_b32") and op.endswith(("_e32", "_dpp", "_sdwa")) and len(ops) == 4:
            pass  # the mask (vcc) is spelled as the fourth operand: already in uses
        if op.startswith(("v_addc", "v_subb", "v_subbrev", "v_add_co", "v_sub_co", "v_subrev_co", "v_mad_u64", "v_mad_i64")):
            return None  # carries: not modelled
    ins.defs, ins.uses = defs, uses
    return ins


# ---------------------------------------------------------------------------------------------
# blocks, dependence graph
# ---------------------------------------------------------------------------------------------
LABEL = re.compile(r"^[.\w$]+:")


def matching_functions(lines, sub):
    return [re.match(r"^(_Z\w+):", l).group(1) for l in lines if re.match(r"^_Z\w+:", l) and sub in l]


def next_free_vgpr(lines, fn):
    inside = False
    for l in lines:
        s_ = l.strip()
        if s_.startswith(".amdhsa_kernel "):
            inside = s_.split()[1] == fn
        elif inside and s_.startswith(".amdhsa_next_free_vgpr"):
            return int(s_.split()[1])
    raise SystemExit(f"no kernel descriptor for {fn}")


def find_function(lines, sub):
    start = end = None
    for i, l in enumerate(lines):
        if start is None and re.match(r"^_Z\w+:", l) and sub in l:
            start = i
        elif start is not None and l.startswith(".Lfunc_end"):
            end = i
            break
    if start is None or end is None:
        raise SystemExit(f"function containing {sub!r} not found")
    return start, end


def straight_pieces(lines, lo, hi):
    """maximal runs of lines [a, b) inside [lo, hi) without labels, branches or anything parse_inst refuses"""
    pieces, cur = [], []
    a = None
    for i in range(lo, hi):
        l = lines[i]
        s = l.strip()
        is_break = False
        ins = None
        if s.startswith(";;#ASM"):
            # inline-asm markers: either an empty fence the sources use against the COMPILER's scheduler, or a wrapper
            # around one ordinary instruction (v_max_f64 / v_min_f64 with an SGPR operand): no machine effect — dropped
            if a is None:
                a = i
            ins = parse_inst("")
            ins.text = None
            cur.append(ins)
            continue
        if LABEL.match(l) or s.startswith("."):
            is_break = True
        else:
            ins = parse_inst(l)
            if ins is None:
                is_break = True
            elif not ins.comment_only and ({"exec_lo", "exec_hi"} & ins.defs):
                is_break = True
        if is_break:
            if cur:
                pieces.append((a, i, cur))
            cur, a = [], None
        else:
            if a is None:
                a = i
            cur.append(ins)
    if cur:
        pieces.append((a, hi, cur))
    return pieces


def build_dag(insts):
    """edges (pred → succ, kind) for RAW / WAR / WAW over registers; LDS reads keep no mutual order (no LDS write in a piece)"""
    n = len(insts)
    preds = [dict() for _ in range(n)]   # pred index → 'raw' | 'war' | 'waw'
    last_def = {}
    last_uses = defaultdict(list)
    for i, ins in enumerate(insts):
        for r in ins.uses:
            if r in last_def:
                preds[i][last_def[r]] = "raw"
        for r in ins.defs:
            if r in last_def and preds[i].get(last_def[r]) != "raw":
                preds[i].setdefault(last_def[r], "waw")
            for u in last_uses[r]:
                if u != i and preds[i].get(u) is None:
                    preds[i][u] = "war"
        for r in ins.defs:
            last_def[r] = i
            last_uses[r] = []
        for r in ins.uses:
            last_uses[r].append(i)
    return preds


# ---------------------------------------------------------------------------------------------
# register renaming inside a piece (round 5): after register allocation the temporaries share so few registers that
# write-after-read / write-after-write dependences pin the compiler's order (round 4: 987 -> 846 modelled ticks with
# the allocated registers, 595 with unlimited renaming).  A kernel variant that is only launched with one wave per
# SIMD may use up to 512 VGPRs, so every VALUE that is born and dies inside a piece gets registers of its own from a
# pool above the function's own.  Same instructions, same operand values, same results bit for bit.
# ---------------------------------------------------------------------------------------------
VREG_TOKEN = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def vregs_in(tok):
    out = []
    for m in VREG_TOKEN.finditer(tok):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def operand_layout(ins):
    """(op, operands, index set of DEF operands) for an instruction the renamer understands, else None"""
    code = ins.text.split(";")[0].rstrip()
    m = re.match(r"(\s*)(\S+)\s*(.*)", code)
    op, rest = m.group(2), m.group(3)
    ops = split_operands(rest)
    if op.startswith("ds_read"):
        return op, ops, {0}
    if op.startswith("v_cmp"):
        return op, ops, set()          # operand 0 is an SGPR pair / vcc
    if op.startswith("v_"):
        return op, ops, {0}
    if op.startswith("s_"):
        return op, ops, set()          # (no VGPR operands)
    return None


def rename_piece(insts, pool_lo, pool_hi):
    """rewrites ins.text / ins.defs / ins.uses in place; returns the number of webs renamed"""
    parent = []
    web_regs = []      # value id -> set of original registers its def covers
    pinned = set()

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    def union(a, b):
        a, b = find(a), find(b)
        if a != b:
            parent[b] = a

    reach = {}         # original VGPR number -> value id now in it (absent: live-in)
    plan = []          # per instruction: None or (op, ops, [per operand: list of (reg, value id or None)])
    for ins in insts:
        if ins.comment_only or ins.op in ("s_waitcnt", "s_nop", ""):
            plan.append(None)
            continue
        lay = operand_layout(ins)
        if lay is None:
            return 0
        op, ops, def_idx = lay
        per = [None] * len(ops)
        tied = bool(FMAC.match(op))
        # uses first (they see the values BEFORE this instruction's definitions)
        for k, tok in enumerate(ops):
            if k in def_idx and not tied:
                continue
            regs = vregs_in(tok)
            ids = [reach.get(r) for r in regs]
            per[k] = list(zip(regs, ids))
            live = [v for v in ids if v is not None]
            for v in live[1:]:
                union(live[0], v)
            if live and len(live) != len(ids):
                pinned.add(live[0])    # a tuple that mixes live-in registers with local values
        for k in def_idx:
            regs = vregs_in(ops[k])
            v = len(parent)
            parent.append(v)
            web_regs.append(set(regs))
            if tied:
                for r, old in per[k]:
                    if old is None:
                        pinned.add(v)
                    else:
                        union(old, v)
            per[k] = [(r, v) for r in regs]
            for r in regs:
                reach[r] = v
        plan.append((op, ops, per))
    for r, v in reach.items():
        pinned.add(v)                  # whatever a register holds at the end of the piece may be read later
    webs = defaultdict(set)
    bad = set()
    for v in range(len(parent)):
        webs[find(v)] |= web_regs[v]
    for v in pinned:
        bad.add(find(v))
    # allocation: every renamed web gets a fresh contiguous range with the parity of its original one
    nxt = pool_lo
    mapping = {}                       # web root -> (orig lo, new lo)
    for root in sorted(webs):
        if root in bad:
            continue
        regs = sorted(webs[root])
        if not regs:
            continue
        lo, hi = regs[0], regs[-1]
        if hi - lo + 1 != len(regs):
            continue
        base = nxt
        if len(regs) > 1 or True:
            if (base ^ lo) & 1:
                base += 1
        if base + len(regs) - 1 > pool_hi:
            continue
        mapping[root] = (lo, base)
        nxt = base + len(regs)
    if not mapping:
        return 0

    def new_reg(r, v):
        if v is None:
            return r
        root = find(v)
        if root not in mapping:
            return r
        lo, base = mapping[root]
        return base + (r - lo)

    for ins, pl in zip(insts, plan):
        if pl is None:
            continue
        op, ops, per = pl
        new_ops = []
        for tok, regs in zip(ops, per):
            if not regs:
                new_ops.append(tok)
                continue
            it = iter(regs)

            def sub(m):
                if m.group(3) is not None:
                    r, v = next(it)
                    return f"v{new_reg(r, v)}"
                a, b = int(m.group(1)), int(m.group(2))
                got = [new_reg(*next(it)) for _ in range(a, b + 1)]
                assert got == list(range(got[0], got[0] + len(got))), (ins.text, got)
                return f"v[{got[0]}:{got[-1]}]"
            new_ops.append(VREG_TOKEN.sub(sub, tok))
        comment = ins.text.split(";", 1)[1] if ";" in ins.text else None
        text = "\t" + op + " " + ", ".join(new_ops)
        ins.text = text
        ins.raw_lines = [text]
        re_parsed = parse_inst(text)
        ins.defs, ins.uses = re_parsed.defs, re_parsed.uses
    return len(mapping)


# ---------------------------------------------------------------------------------------------
# in-order issue simulation of one wave (no other wave on the SIMD)
# ---------------------------------------------------------------------------------------------
def hazard_gap(prod, cons):
    """wait states the gfx940 family needs between producer and consumer (0 = none): GCNHazardRecognizer's rules that can
    occur in the pieces this tool touches"""
    need = 0
    if prod.is_valu:
        sg = {r for r in prod.defs if r[0] == "s" or r.startswith("vcc")}
        if sg and cons.is_valu and (sg & cons.uses):
            need = max(need, 2)          # VALU writes SGPR / VCC → VALU reads it (mask or constant)
        if sg and cons.op.startswith("v_readlane") and (sg & cons.uses):
            need = max(need, 4)
        if prod.is_trans and cons.is_valu and not cons.is_trans and (prod.defs & cons.uses):
            need = max(need, 1)          # trans result forwarded to a non-trans VALU
        if cons.op.startswith("v_readlane") and ({r for r in prod.defs if r[0] == "v"} & cons.uses):
            need = max(need, 1)
    if prod.op.startswith("s_") and ("m0" in prod.defs) and ("m0" in cons.uses):
        need = max(need, 1)
    return need


def simulate(order, insts, preds, lds_latency=None):
    """cycles until the last result of the piece is available, issuing `order` in order on an otherwise idle SIMD"""
    ready = {}
    t = 0
    finish = 0
    lat_lds = lds_latency or MODEL["lds"][1]
    for pos, i in enumerate(order):
        ins = insts[i]
        if ins.comment_only or ins.op == "s_waitcnt":
            continue
        if ins.op == "s_nop":
            continue
        issue, lat = MODEL[ins.kind]
        if ins.is_lds:
            lat = lat_lds
        start = t
        for p, kind in preds[i].items():
            if kind == "raw":
                start = max(start, ready.get(p, 0))
        ready[i] = start + lat
        t = start + issue
        finish = max(finish, ready[i])
    return finish


def list_schedule(insts, preds):
    """critical-path list scheduling with the measured latencies, in-order issue: at every step take, among the
    instructions whose predecessors have been scheduled, the one that can start earliest; ties by longest path to the end"""
    n = len(insts)
    succs = [[] for _ in range(n)]
    for i in range(n):
        for p, k in preds[i].items():
            succs[p].append((i, k))
    # priority: longest latency-weighted path to the end of the piece
    height = [0] * n
    for i in range(n - 1, -1, -1):
        ins = insts[i]
        issue, lat = MODEL[ins.kind] if not (ins.comment_only or ins.op in ("s_waitcnt", "s_nop")) else (0, 0)
        h = lat
        for s, k in succs[i]:
            h = max(h, (lat if k == "raw" else issue) + height[s])
        height[i] = h
    indeg = [len(preds[i]) for i in range(n)]
    avail = [i for i in range(n) if indeg[i] == 0]
    ready = {}
    t = 0
    order = []
    while avail:
        best, best_key = None, None
        for i in avail:
            ins = insts[i]
            st = t
            for p, kind in preds[i].items():
                if kind == "raw":
                    st = max(st, ready.get(p, 0))
            key = (max(st, t), -height[i], i)
            if best_key is None or key < best_key:
                best, best_key = i, key
        i = best
        avail.remove(i)
        ins = insts[i]
        if ins.comment_only or ins.op in ("s_waitcnt", "s_nop"):
            issue, lat = 0, 0
        else:
            issue, lat = MODEL[ins.kind]
        st = best_key[0]
        ready[i] = st + lat
        t = st + issue
        order.append(i)
        for s, k in succs[i]:
            indeg[s] -= 1
            if indeg[s] == 0:
                avail.append(s)
    return order


# ---------------------------------------------------------------------------------------------
# emission: waitcnts for the LDS reads in their new order, hazard no-ops
# ---------------------------------------------------------------------------------------------
def emit(order, insts, preds, entry_waitcnt):
    """`entry_waitcnt`: lgkmcnt(0) is emitted ahead of the piece's first LDS-dependent use only as the reads require;
    reads issued BEFORE the piece are not known here, so a piece that had a waitcnt at its head keeps it (entry_waitcnt)."""
    out = []
    lds_issued = []       # instruction indices of the LDS reads issued so far in the piece, in order
    waited_upto = 0       # reads [0, waited_upto) are known to have landed
    emitted = []          # Inst objects in emission order (for hazard distances)
    if entry_waitcnt:
        out.append(entry_waitcnt)
    for i in order:
        ins = insts[i]
        if ins.comment_only:
            if ins.text is not None:
                out.append(ins.text)
            continue
        if ins.op in ("s_waitcnt", "s_nop"):
            continue      # recomputed
        # LDS results this instruction needs (RAW) — or overwrites / re-reads as WAR/WAW on an in-flight destination
        need = -1
        for p, kind in preds[i].items():
            if insts[p].is_lds and p in lds_issued:
                need = max(need, lds_issued.index(p))
        if need >= waited_upto:
            outstanding_allowed = min(15, len(lds_issued) - 1 - need)   # (the counter has four bits)
            out.append(f"\ts_waitcnt lgkmcnt({outstanding_allowed})")
            emitted.append(None)
            waited_upto = need + 1
        # hazards against the previous few emitted instructions
        gap_needed = 0
        dist = 0
        for prev in reversed(emitted[-6:]):
            if prev is None:      # a waitcnt counts as a wait state
                dist += 1
                continue
            g = hazard_gap(prev, ins)
            if g > dist:
                gap_needed = max(gap_needed, g - dist)
            dist += 1
        if gap_needed > 0:
            out.append(f"\ts_nop {gap_needed - 1}")
            for _ in range(gap_needed):
                emitted.append(None)
        out.append(ins.text)
        emitted.append(ins)
        if ins.is_lds:
            lds_issued.append(i)
    # everything this piece started must be assumed pending by whoever follows: the following code's own waitcnts were
    # computed by the compiler for ITS order of reads — a piece that ends with reads in flight ends with a full wait
    if waited_upto < len(lds_issued):
        out.append("\ts_waitcnt lgkmcnt(0)")
    return out


# ---------------------------------------------------------------------------------------------
# verification: the emitted piece against the compiler's, by symbolic execution (run on every edit; a failure stops the build)
# ---------------------------------------------------------------------------------------------
ANY_REG = re.compile(r"\b([vsa])\[(\d+):(\d+)\]|\b([vs])(\d+)\b|\b(vcc_lo|vcc_hi|vcc|exec_lo|exec_hi|exec|m0|scc)\b")


def _expand(m):
    if m.group(1):
        return [f"{m.group(1)}{n}" for n in range(int(m.group(2)), int(m.group(3)) + 1)]
    if m.group(4):
        return [f"{m.group(4)}{m.group(5)}"]
    r = m.group(6)
    return {"vcc": ["vcc_lo", "vcc_hi"], "exec": ["exec_lo", "exec_hi"]}.get(r, [r])


def symbolic_state(raw_lines, scratch_lo=None, scratch_hi=None):
    """Runs a straight-line piece on symbolic inputs.  Every instruction's result is a hash of its mnemonic, its modifiers and
    the VALUES (not the names) of what it reads, operand by operand; returns (final value of every register written, the
    multiset of instructions by value).  Also checks the LDS discipline of the piece's own reads: a result is used, and a
    destination overwritten, only behind an s_waitcnt lgkmcnt that covers it."""
    val = {}
    get = lambda r: val.get(r, ("in", r))
    bag = []
    issued = 0            # LDS reads of this piece issued so far
    landed = 0            # reads [0, landed) are known to have arrived
    pending = {}          # register -> index of the in-flight read that writes it
    for line in raw_lines:
        if line is None:
            continue
        code = line.split(";")[0].strip()
        if not code or code.startswith(";;#") or code.startswith("."):
            continue
        m = re.match(r"(\S+)\s*(.*)", code)
        op, rest = m.group(1), m.group(2)
        if op == "s_nop":
            continue
        if op == "s_waitcnt":
            c = re.search(r"lgkmcnt\((\d+)\)", rest)
            if c:
                landed = max(landed, issued - int(c.group(1)))
                pending = {r: i for r, i in pending.items() if i >= landed}
            continue
        ops = split_operands(rest)
        ins = parse_inst(line)
        if ins is None:
            raise SystemExit(f"verify: cannot parse {line!r}")
        if op.startswith("ds_read") or (op.startswith("v_") and not op.startswith("v_cmp")) or (op.startswith("s_") and not op.startswith(("s_cmp", "s_bitcmp"))):
            def_idx = {0}
        elif op.startswith("v_cmp"):
            def_idx = {0}
        else:
            def_idx = set()
        tied = bool(FMAC.match(op))
        reads = []
        canon = [op]
        for k, tok in enumerate(ops):
            if k in def_idx and not tied:
                canon.append("D")
                continue
            parts = []

            def sub(mm):
                rs = _expand(mm)
                reads.extend(rs)
                parts.append(tuple(get(r) for r in rs))
                return "@"
            shape = ANY_REG.sub(sub, tok)
            canon.append((shape, tuple(parts)))
        implicit = sorted(ins.uses - set(reads))      # exec for VALU, scc for s_addc …
        for r in implicit:
            reads.append(r)
            canon.append((r, get(r)))
        for r in reads:
            if r in pending:
                raise SystemExit(f"verify: {line.strip()!r} reads {r} while the LDS read that writes it may be in flight")
        for r in ins.defs:
            if r in pending:
                raise SystemExit(f"verify: {line.strip()!r} overwrites {r}, the destination of an LDS read in flight")
        h = hash(tuple(canon))
        bag.append(h)
        defs_in_order = []
        for k in sorted(def_idx):
            if k < len(ops):
                for mm in ANY_REG.finditer(ops[k]):
                    defs_in_order.extend(_expand(mm))
        for r in sorted(ins.defs - set(defs_in_order)):
            defs_in_order.append(r)                    # implicit: scc of an s_and, exec of a saveexec (not in pieces)
        for pos, r in enumerate(defs_in_order):
            val[r] = (h, pos)
        if op.startswith("ds_read"):
            for r in defs_in_order:
                pending[r] = issued
            issued += 1
    scratch = set()
    if scratch_lo is not None:
        scratch = {f"v{n}" for n in range(scratch_lo, scratch_hi + 1)}
    return {r: v for r, v in val.items() if r not in scratch}, sorted(bag)


def verify_piece(old_lines, new_lines, pool):
    lo, hi = pool if pool else (None, None)
    a_state, a_bag = symbolic_state(old_lines)
    b_state, b_bag = symbolic_state(new_lines, lo, hi)
    if a_bag != b_bag:
        raise SystemExit("verify: the emitted piece does not execute the same instructions on the same values")
    for r, v in a_state.items():
        if b_state.get(r) != v:
            raise SystemExit(f"verify: register {r} ends the piece with another value")
    extra = set(b_state) - set(a_state)
    if extra:
        raise SystemExit(f"verify: the emitted piece writes registers the original does not: {sorted(extra)[:8]}")


NO_REORDER = False


def process(lines, fn_sub, must_contain, report, min_len=12, pool=None):
    lo, hi = find_function(lines, fn_sub)
    pieces = straight_pieces(lines, lo, hi)
    new_lines = list(lines)
    edits = []
    for a, b, insts in pieces:
        real = [x for x in insts if not x.comment_only and x.op not in ("s_waitcnt", "s_nop")]
        if len(real) < min_len:
            continue
        if must_contain and not any(must_contain in x.op for x in real):
            continue
        # a waitcnt that is not lgkmcnt-only (vmcnt / expcnt) pins the piece: leave it alone
        if any(x.op == "s_waitcnt" and not re.fullmatch(r"\s*s_waitcnt lgkmcnt\(\d+\)\s*", x.text.split(";")[0]) for x in insts):
            continue
        # reads in flight at the head of the piece: the compiler's first waitcnt tells; keep a full wait at the head if the
        # original piece waited before its first own read was issued
        # Loads issued BEFORE the piece (LDS reads, scalar loads — both count in lgkmcnt, scalar ones return out of order) that
        # the piece consumes are covered by waitcnts somewhere inside the original piece; the recomputed waitcnts only know the
        # piece's own LDS reads.  So: a piece that waited at all starts with a full wait.  (A piece that never waited consumes
        # nothing that was still in flight at its head.)
        entry = "\ts_waitcnt lgkmcnt(0)" if any(x.op == "s_waitcnt" for x in insts) else None
        preds = build_dag(insts)
        base_order = list(range(len(insts)))
        before = simulate(base_order, insts, preds)
        renamed = 0
        if pool:
            renamed = rename_piece(insts, pool[0], pool[1])
            if renamed:
                preds = build_dag(insts)
        order = base_order if NO_REORDER else list_schedule(insts, preds)
        after = simulate(order, insts, preds)
        if after >= before and not (renamed and NO_REORDER):
            continue
        text = emit(order, insts, preds, entry)
        verify_piece(lines[a:b], text, pool)
        edits.append((a, b, text, before, after, len(real), renamed))
    for a, b, text, before, after, nreal, renamed in sorted(edits, reverse=True):
        new_lines[a:b] = text
        if report:
            print(f"  lines {a}-{b}: {nreal} instructions, modelled lone-wave cycles {before} -> {after} ({renamed} values renamed)", file=sys.stderr)
    return new_lines, edits


def set_vgpr_count(lines, fn_sub, n):
    """the .amdhsa_kernel block and the metadata entry of every kernel whose name contains fn_sub.

    Raising next_free_vgpr / accum_offset to n hands the kernel the registers [old next_free_vgpr, n) as the rename pool.  That
    is only sound for a kernel that uses NO accumulation registers (they would start at the old accum_offset) and NO scratch
    (a spill slot's register may be live where the pool says free): both are checked here, per kernel, in the descriptor and
    in the metadata, and a violation stops the build."""
    out = list(lines)
    in_desc = False
    in_meta = False
    meta_agpr = None   # (.agpr_count precedes .name in a metadata entry: remembered until the name says whose it is)
    for i, l in enumerate(out):
        s_ = l.strip()
        if s_.startswith(".amdhsa_kernel "):
            in_desc = fn_sub in s_
        elif s_.startswith(".end_amdhsa_kernel"):
            in_desc = False
        elif in_desc and s_.startswith(".amdhsa_next_free_vgpr"):
            out[i] = f"\t\t.amdhsa_next_free_vgpr {n}"
        elif in_desc and s_.startswith(".amdhsa_accum_offset"):
            out[i] = f"\t\t.amdhsa_accum_offset {n}"
        elif in_desc and s_.startswith(".amdhsa_private_segment_fixed_size"):
            if int(s_.split()[-1]) != 0:
                raise SystemExit(f"gcn_sched: a kernel matching {fn_sub!r} uses scratch ({s_}): its registers cannot be renamed into a pool")
        if s_.startswith("- .agpr_count:"):
            meta_agpr = int(s_.split()[-1])
        if s_.startswith(".name:"):
            in_meta = fn_sub in s_
            if in_meta and meta_agpr not in (None, 0):
                raise SystemExit(f"gcn_sched: kernel {s_.split()[-1]} uses {meta_agpr} AGPRs: raising accum_offset to {n} would move them")
        if in_meta and s_.startswith(".private_segment_fixed_size:") and int(s_.split()[-1]) != 0:
            raise SystemExit(f"gcn_sched: a kernel matching {fn_sub!r} uses scratch ({s_})")
        if s_.startswith(".vgpr_count:"):
            # metadata entries list .name before .vgpr_count (alphabetical keys: .name < .vgpr_count)
            if in_meta:
                out[i] = re.sub(r"\d+", str(n), l)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--function", action="append", required=True)
    ap.add_argument("--blocks-with", default="")
    ap.add_argument("--report", action="store_true")
    ap.add_argument("--rename", default="", help="LO:HI — VGPRs the function does not use, for values local to a piece (LO = auto: the kernel's own next_free_vgpr)")
    ap.add_argument("--no-reorder", action="store_true", help="rename only: the compiler's order (A/B)")
    ap.add_argument("--latency-scale", type=float, default=1.0, help="multiplies the model's VALU result latencies (A/B)")
    ap.add_argument("--all-matching", action="store_true", help="every function whose name contains a --function string")
    ap.add_argument("--vgprs", type=int, default=0, help="rewrite the kernel descriptors' VGPR counts (next_free_vgpr, accum_offset, .vgpr_count)")
    a = ap.parse_args()
    if a.latency_scale != 1.0:
        for k, (iss, lat) in list(MODEL.items()):
            if k not in ("lds", "nop", "salu"):
                MODEL[k] = (iss, int(round(lat * a.latency_scale)))
    globals()["NO_REORDER"] = a.no_reorder
    lines = open(a.src).read().split("\n")
    fns = a.function
    if a.all_matching:
        fns = [f for sub in a.function for f in matching_functions(lines, sub)]
        if not fns:
            raise SystemExit(f"no function matches {a.function}")
    for fn in fns:
        if a.report:
            print(fn, file=sys.stderr)
        pool = None
        if a.rename:
            lo, hi = a.rename.split(":")
            lo = next_free_vgpr(lines, fn) if lo == "auto" else int(lo)
            lo += lo & 1
            pool = (lo, int(hi))
            if a.report:
                print(f"  rename pool v{lo}..v{hi}", file=sys.stderr)
        lines, _ = process(lines, fn, a.blocks_with, a.report, pool=pool)
        if a.vgprs:
            lines = set_vgpr_count(lines, fn, a.vgprs)
    open(a.dst, "w").write("\n".join(lines))
