#!/usr/bin/env python3
"""Summarise the rocprofv3 counter passes written by tools/pmc_run.sh: per kernel (name substring), averages per launch of every counter
found under <dir>/<tag>_<pass>/**/**counter_collection.csv, plus derived lines (per-wave instruction counts, VALU busy, wait fractions,
LDS bank conflicts, gfx950-corrected HBM bytes).  usage: python tools/pmc_summary.py <dir> <tag> <kernel substring> [--json out.json]"""
import csv, glob, json, os, sys

src, tag, match = sys.argv[1], sys.argv[2], sys.argv[3]
jout = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
tot, name = {}, None
files = glob.glob(os.path.join(src, tag + "_*", "**", "*counter_collection.csv"), recursive=True)
# only the launches of the benchmark's own corpus: the largest grid the kernel was launched with (the parity probe of bench.py
# launches the same kernels on a 200-400 document sample)
gmax = 0
for f in files:
    for r in csv.DictReader(open(f)):
        if match in r["Kernel_Name"]:
            gmax = max(gmax, int(r["Grid_Size"]))
for f in files:
    per = {}
    for r in csv.DictReader(open(f)):
        if match not in r["Kernel_Name"] or int(r["Grid_Size"]) != gmax:
            continue
        name = r["Kernel_Name"]
        key = (r["Counter_Name"], r["Dispatch_Id"])
        per[key] = per.get(key, 0.0) + float(r["Counter_Value"])       # one row per XCD / instance: sum them
    for (c, _), v in per.items():
        t = tot.setdefault(c, [0.0, 0]); t[0] += v; t[1] += 1
avg = {c: v[0] / v[1] for c, v in tot.items()}
print("kernel: %s   (launches with grid size %d only)" % (name, gmax))
print("launches averaged: %s" % {c: v[1] for c, v in tot.items()})
for c in sorted(avg):
    print("  %-28s %16.0f" % (c, avg[c]))
d = {}
if "SQ_WAVES" in avg and avg["SQ_WAVES"]:
    w = avg["SQ_WAVES"]
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
        if c in avg:
            d[c + "_per_wave"] = avg[c] / w
if "SQ_WAVE_CYCLES" in avg:
    wc = avg["SQ_WAVE_CYCLES"]
    for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS"):
        if c in avg:
            d[c + "_over_WAVE_CYCLES"] = avg[c] / wc
if "SQ_LDS_BANK_CONFLICT" in avg and avg.get("SQ_LDS_IDX_ACTIVE"):
    d["LDS_bank_conflict_over_LDS_active"] = avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_LDS_IDX_ACTIVE"]
if "SQ_ACTIVE_INST_VALU" in avg and avg.get("SQ_BUSY_CYCLES"):
    # SQ_ACTIVE_INST_VALU counts quad-cycles summed over waves; SQ_BUSY_CYCLES is summed over the SQs (one per CU... per XCD instance rows)
    d["VALU_active_quadcycles_per_SQ_busy_cycle"] = avg["SQ_ACTIVE_INST_VALU"] / avg["SQ_BUSY_CYCLES"]
if "FETCH_SIZE" in avg or "WRITE_SIZE" in avg:
    f, wr = avg.get("FETCH_SIZE", 0.0), avg.get("WRITE_SIZE", 0.0)
    d["hbm_bytes_per_launch_gfx950_corrected"] = (2.0 * f + wr) * 1024.0
    d["FETCH_SIZE_KB"] = f; d["WRITE_SIZE_KB"] = wr
print("derived:")
for k in sorted(d):
    print("  %-44s %14.4f" % (k, d[k]))
if jout:
    # provenance: which sources the profiled library was built from -- bench.py attaches the counters only to a build with the same hash
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    json.dump({"kernel": name, "csrc_sha16": bench.csrc_sha16("lda" if "k_lda" in match else "ctm"), "counters_per_launch": avg, "derived": d,
               "hbm_bytes_per_launch_gfx950_corrected": d.get("hbm_bytes_per_launch_gfx950_corrected"),
               "note": "rocprofv3 --pmc, one pass per counter set (tools/pmc_run.sh); FETCH_SIZE counts 64 B per 128-B request on gfx950 -> x2 on the "
                       "read side (MI355X_MICROARCH.md); fabric-side counters: Infinity-Cache hits are included"}, open(jout, "w"), indent=1)
