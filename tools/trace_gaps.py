#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV: the kernels of N consecutive LDA iterations in the middle of the run, start / end relative to the first,
duration and the gap to the previous kernel's end.   usage: trace_gaps.py <kernel_trace.csv> [iterations=4] [skip_from_end=400]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 400
short = lambda s: s.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Grid_Size_X"])) for r in rows), key=lambda e: e[0])
gmax = max(e[3] for e in ev if e[2].startswith("k_lda_estep"))
idx = [i for i, e in enumerate(ev) if e[2].startswith("k_lda_estep") and e[3] == gmax]
i0 = idx[max(0, len(idx) - skip)]
sel = []
for e in ev[i0:]:
    sel.append(e)
    if sum(1 for x in sel if x[2].startswith("k_lda_estep")) > n:
        break
t0 = sel[0][0]
prev = None
print("%-60s %9s %9s %8s %8s" % ("kernel", "start us", "end us", "dur us", "gap us"))
for s, e, name, g in sel[:-1]:
    print("%-60s %9.2f %9.2f %8.2f %8s" % (name[:60], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, "" if prev is None else "%.2f" % ((s - prev) / 1e3)))
    prev = e
