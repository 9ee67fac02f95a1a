#!/usr/bin/env python3
"""Where a kernel waits for memory: the vector-memory instructions, scratch traffic, vmcnt waits and barriers of one kernel in a
hipcc -S listing, with their line offsets and the loop headers between them.  A load requested a step ahead only stays in flight if
no vmcnt wait (e.g. for a scratch reload, or a wait priced as vmcnt(0) because of lane-conditional stores) sits between request and use:
vmcnt counts in order.   usage: asm_vmem_waits.py file.s kernel-substring [first_line last_line]"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(sys.argv[2]) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, len(body))
for i, l in enumerate(body):
    if lo <= i < hi and re.search(r"scratch_|global_|buffer_|flat_|vmcnt|s_barrier|Loop Header", l):
        print(i, l.strip()[:130])
