#!/usr/bin/env python3
"""Instruction budget of the two LD_MMA loops of a solve kernel in a hipcc -S listing, by class -- the table of DESIGN section 4.2.
Per depth-1 loop: the COMMON trip (the loop's blocks that every trip executes = up to the first conditional block) and the conditional
blocks (rho growth, x-tolerance test, slot refill), each as
  f64 arithmetic (add / mul / fma / max / min / ldexp / rndne / cvt, and rcp / rsq / sqrt at 16 issue cycles) -- what NLopt's formulas and the
  objectives prescribe, plus the table-driven exp / log;
  lane sums (DPP moves + the adds that go with them are under f64);
  state machine (v_cndmask, v_cmp*, v_and/or on masks);
  moves (v_mov*, v_accvgpr*, v_readlane / readfirstlane);
  integer / address (v_*_u32, v_*_i32, v_lshl*, v_mad_u64 ...);
  LDS, memory, scalar.
usage: asm_trip_budget.py file.s kernel-substring"""
import re
import sys

src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(key) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]

CLASSES = ["f64", "trans", "dpp", "state", "move", "int", "lds", "vmem", "salu", "other"]


def classify(l):
    t = l.strip().split()
    if not t or t[0].startswith((";", ".")) or t[0].endswith(":"):
        return None
    op = t[0]
    if re.match(r"v_(rcp|rsq|sqrt)_f64", op):
        return "trans"
    if op.startswith("v_"):
        if "dpp" in l:
            return "dpp"
        if re.match(r"v_(add|mul|fma|fmac|max|min|ldexp|rndne|trunc|floor|fract|div_scale|div_fmas|div_fixup)_f64", op) or re.match(r"v_cvt_(f64_i32|i32_f64|f64_u32|f64_f32|f32_f64)", op):
            return "f64"
        if re.match(r"v_(cndmask|cmp|cmpx)", op):
            return "state"
        if re.match(r"v_(mov|accvgpr|readlane|readfirstlane|writelane|swap|permlane)", op):
            return "move"
        if re.match(r"v_(mul|add|sub|max|min|fma)_f32", op):
            return "state"          # the sign bookkeeping of sigma (floats)
        return "int"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def tally(ls):
    n = {c: 0 for c in CLASSES}
    for l in ls:
        c = classify(l)
        if c:
            n[c] += 1
    n["vector_issue_cycles"] = 4 * (n["f64"] + n["dpp"] + n["state"] + n["move"] + n["int"]) + 16 * n["trans"]
    return n


hdrs = [(i, re.match(r"^(\.LBB\d+_\d+):", l).group(1)) for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:.*Loop Header: Depth=1", l)]
for i, h in hdrs:
    name = h[2:]
    last = i
    for j in range(i, len(body)):
        if re.match(r"^\.LBB\d+_\d+:", body[j]) and ("Header=" + name + " ") in body[j]:
            last = j
    k = last + 1
    while k < len(body) and not re.match(r"^\.LBB\d+_\d+:", body[k]):
        k += 1
    loop = body[i:k]
    if tally(loop)["f64"] < 40:
        continue
    # blocks of the loop
    blocks, cur = [], None
    for l in loop:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = [m.group(1), []]; blocks.append(cur)
        elif cur is not None:
            cur[1].append(l)
    common = tally(blocks[0][1])
    cond = tally([l for b in blocks[1:] for l in b[1]])
    print("loop %s: %d blocks" % (h, len(blocks)))
    print("   common trip      ", {c: common[c] for c in CLASSES if common[c]}, "vector issue cycles", common["vector_issue_cycles"])
    print("   conditional parts", {c: cond[c] for c in CLASSES if cond[c]}, "vector issue cycles", cond["vector_issue_cycles"])
