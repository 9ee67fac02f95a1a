#!/usr/bin/env python3
"""The basic blocks of a kernel's big depth-1 loops in a hipcc -S listing: per block the vector instructions (f64 / selects and compares / other),
memory and scalar instructions and the branches that leave it -- to see WHICH conditional block of the LD_MMA state machine carries what.
usage: asm_loop_blocks.py file.s kernel-substring [min-f64-instructions-per-loop=100]"""
import re
import sys

src, key = sys.argv[1], sys.argv[2]
minf = int(sys.argv[3]) if len(sys.argv) > 3 else 100
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(key) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]


def op(l):
    t = l.strip().split()
    if not t or t[0].startswith((";", ".")) or t[0].endswith(":"):
        return None
    return t[0]


for hi in [i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:.*Loop Header: Depth=1", l)]:
    name = re.match(r"^\.L(BB\d+_\d+):", body[hi]).group(1)
    last = hi
    for j in range(hi, len(body)):
        if re.match(r"^\.LBB\d+_\d+:", body[j]) and ("Header=" + name + " ") in body[j]:
            last = j
    k = last + 1
    while k < len(body) and not re.match(r"^\.LBB\d+_\d+:", body[k]):
        k += 1
    loop = body[hi:k]
    ops = [o for o in map(op, loop) if o]
    if sum(o.endswith("_f64") for o in ops) < minf:
        continue
    print("loop %s: %d instructions" % (name, len(ops)))
    blocks, cur = [], None
    for l in loop:
        m = re.match(r"^\.L(BB\d+_\d+):", l)
        if m:
            cur = [m.group(1), []]; blocks.append(cur)
        elif cur is not None and op(l):
            cur[1].append(l.strip())
    for n, ls in blocks:
        o = [x.split()[0] for x in ls]
        v = [x for x in o if x.startswith("v_")]
        f64 = sum(x.endswith("_f64") for x in v); cnd = sum(x.startswith(("v_cndmask", "v_cmp")) for x in v)
        print("  %-10s valu %4d (f64 %3d, select/compare %3d, other %3d)  lds %2d vmem %2d salu %3d  %s" % (
            n, len(v), f64, cnd, len(v) - f64 - cnd, sum(x.startswith("ds_") for x in o), sum(x.startswith(("global_", "buffer_", "flat_", "scratch_")) for x in o),
            sum(x.startswith("s_") for x in o), " ".join(x for x in ls if x.startswith("s_cbranch"))))
