#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes of `bench.py` (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, separate runs as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) into profiles/r01_traffic_lda_estep.json, which bench.py reports as
roofline.traffic.  Usage: python tools/collect_traffic.py <dir with *_counter_collection.csv of both passes> <out.json> [kernel substring] [config text]"""
import csv, glob, json, os, sys

src, out = sys.argv[1], sys.argv[2]
match = sys.argv[3] if len(sys.argv) > 3 else "k_lda_estep"       # kernel-name substring
config = sys.argv[4] if len(sys.argv) > 4 else "bench.py default: LDA K=10, 10000 docs x 96 terms, 1 GPU"
tot = {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]}
name = None
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    per = {}
    for r in csv.DictReader(open(f)):
        if match not in r["Kernel_Name"] or r["Counter_Name"] not in tot:
            continue
        name = r["Kernel_Name"]
        key = (r["Counter_Name"], r["Dispatch_Id"])
        per[key] = per.get(key, 0.0) + float(r["Counter_Value"])       # one row per XCD / instance: sum them
    for (c, _), v in per.items():
        tot[c][0] += v; tot[c][1] += 1
fetch_kb = tot["FETCH_SIZE"][0] / max(tot["FETCH_SIZE"][1], 1)
write_kb = tot["WRITE_SIZE"][0] / max(tot["WRITE_SIZE"][1], 1)
res = {"kernel": name, "config": config,
       "launches_averaged": [tot["FETCH_SIZE"][1], tot["WRITE_SIZE"][1]],
       "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
       "hbm_bytes_per_launch_raw": (fetch_kb + write_kb) * 1024.0,
       "hbm_bytes_per_launch_gfx950_corrected": (2.0 * fetch_kb + write_kb) * 1024.0,
       "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH_SIZE counts 64 B per 128-B request on gfx950 -> x2 "
               "on the read side (MI355X_MICROARCH.md).  Fabric-side counters: Infinity-Cache hits are included."}
if match == "k_lda_estep":
    res["note"] += ("  The 11 MB working set is Infinity-Cache resident.  Writes: 1.6 MB Elntheta + gamma_next and the per-block lambda "
                    "partials (grid x 7,680 B).")
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
