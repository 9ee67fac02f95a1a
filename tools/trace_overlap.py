#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV: for the last N passes of a CTM run, the start / end of every kernel (relative to the pass's first
kernel) with its queue, and for each `k_reduce_partials` launch the kernels it overlaps in time -- does the side-stream reduction of the gamma
statistics run BESIDE the solve kernel or delay it?   usage: trace_overlap.py <kernel_trace.csv> [passes=3] [skip_last=0]
(skip_last: bench.py's final repeat records per-phase event spans, during which the library keeps everything in stream order)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
npass = int(sys.argv[2]) if len(sys.argv) > 2 else 3
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"), int(r["Grid_Size_X"])) for r in rows), key=lambda e: e[0])
# the passes over the benchmark's own corpus: the solve launches with the largest grid (the parity probe fits a 400-document sample)
gmax = max(e[4] for e in ev if e[2].startswith("k_ctm_solve"))
solves = [i for i, e in enumerate(ev) if e[2].startswith("k_ctm_solve") and e[4] == gmax]
if skip:
    solves = solves[:-skip]
first = solves[-npass] if len(solves) >= npass else solves[0]
stop = solves[-1]
while stop + 1 < len(ev) and not ev[stop][2].startswith(("k_ctm_loglik", "k_ll_")):
    stop += 1
ev = [e[:4] for e in ev[:stop + 2]]
# a pass starts at the theta launch(es) before its solve: walk back to the previous log-likelihood kernel
start = first
while start > 0 and not ev[start - 1][2].startswith(("k_ctm_loglik", "k_ll_")):
    start -= 1
t0 = ev[start][0]
print("%-58s %6s %10s %10s %9s" % ("kernel", "queue", "start us", "end us", "dur us"))
for s, e, n, q in ev[start:]:
    print("%-58s %6s %10.1f %10.1f %9.1f" % (n[:58], q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
print()
for s, e, n, q in ev[start:]:
    if not n.startswith("k_reduce_partials"):
        continue
    over = [(n2, q2, max(s, s2), min(e, e2)) for s2, e2, n2, q2 in ev[start:] if (s2, e2, n2) != (s, e, n) and s2 < e and e2 > s]
    print("k_reduce_partials on queue %s, %.1f us (%.1f .. %.1f): overlaps %s" % (q, (e - s) / 1e3, (s - t0) / 1e3, (e - t0) / 1e3,
          ", ".join("%s [queue %s] for %.1f us" % (n2[:40], q2, (b - a) / 1e3) for n2, q2, a, b in over) or "nothing"))
