#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of a CTM run: every kernel of N consecutive passes near the end of the run (a pass = from one theta-phase
launch to the next), start / end relative to the first, duration, queue, and the gap to the latest end seen so far.
usage: trace_ctm_pass.py <kernel_trace.csv> [passes=2] [skip_from_end=6]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 6
short = lambda s: s.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"), int(r["Grid_Size_X"])) for r in rows), key=lambda e: e[0])
solves = [i for i, e in enumerate(ev) if e[2].startswith("k_ctm_solve")]
first = lambda i: max(j for j in range(i + 1) if ev[j][2].startswith(("k_ctm_theta", "k_ctm_estep")) and (j == 0 or not ev[j - 1][2].startswith(("k_ctm_theta", "k_ctm_estep"))))
i0 = first(solves[-skip - n])
i1 = first(solves[-skip])
t0 = ev[i0][0]
latest = None
print("%-58s %5s %9s %9s %8s %8s" % ("kernel", "queue", "start us", "end us", "dur us", "gap us"))
for s, e, name, q, g in ev[i0:i1]:
    print("%-58s %5s %9.2f %9.2f %8.2f %8s" % (name[:58], q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, "" if latest is None else "%.2f" % ((s - latest) / 1e3)))
    latest = e if latest is None else max(latest, e)
print("%d passes: %.2f us per pass" % (n, (ev[i1][0] - t0) / 1e3 / n))
