#!/usr/bin/env python3
"""Static VALU issue cost of the loops of one kernel in a hipcc -S listing: per top-level loop, the number of vector instructions and
their issue cycles on gfx950 (measured, profiles/experiments/r03_f64_rates.txt: 4 cycles per wave instruction, 16 for v_rcp/rsq/sqrt_f64).
usage: asm_loop_cost.py file.s kernel-substring [first_label last_label]   (labels like .LBB30_63; default: every depth-1 loop)"""
import re, sys
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(key) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
def cost(ls):
    n = {"valu": 0, "trans": 0, "lds": 0, "salu": 0, "vmem": 0, "dpp": 0, "cnd": 0}
    for l in ls:
        t = l.strip().split()
        if not t or t[0].startswith((";", ".")) or t[0].endswith(":"): continue
        op = t[0]
        if re.match(r"v_(rcp|rsq|sqrt)_f64", op): n["trans"] += 1
        elif op.startswith("v_"):
            n["valu"] += 1
            if "dpp" in l: n["dpp"] += 1
            if op.startswith("v_cndmask"): n["cnd"] += 1
        elif op.startswith("ds_"): n["lds"] += 1
        elif op.startswith("s_"): n["salu"] += 1
        elif op.startswith(("global_", "scratch_", "buffer_", "flat_")): n["vmem"] += 1
    n["valu_cycles"] = 4 * n["valu"] + 16 * n["trans"]
    return n
if len(sys.argv) > 4:
    a = next(i for i, l in enumerate(body) if l.startswith(sys.argv[3] + ":"))
    b = next(i for i, l in enumerate(body) if l.startswith(sys.argv[4] + ":"))
    print(sys.argv[3], "..", sys.argv[4], cost(body[a:b]))
else:
    hdrs = [(i, re.match(r"^(\.LBB\d+_\d+):", l).group(1)) for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:.*Loop Header: Depth=1", l)]
    for i, h in hdrs:
        name = h[2:]
        last = i
        for j in range(i, len(body)):
            if re.match(r"^\.LBB\d+_\d+:", body[j]) and ("Header=" + name + " ") in body[j]: last = j
        k = last + 1
        while k < len(body) and not re.match(r"^\.LBB\d+_\d+:", body[k]): k += 1
        c = cost(body[i:k])
        if c["valu"] + c["trans"] > 40: print(h, "lines", k - i, c)
    print("whole kernel", cost(body))
