#!/usr/bin/env python3
"""Copies the summaries tools/refresh_evidence.sh left under gpurun_out/evidence/ into profiles/ under their round-4 names."""
import glob, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E, P = os.path.join(ROOT, "gpurun_out", "evidence"), os.path.join(ROOT, "profiles")
pairs = {"bench_default.json": "r04_final_bench_default.json", "bench_cfg4.json": "r04_final_bench_cfg4.json", "bench_cfg5.json": "r04_final_bench_cfg5.json",
         "bench_cfg2_steps20.json": "r04_final_bench_cfg2_steps20.json", "lda_scaling.jsonl": "r04_lda_scaling.jsonl",
         "lda_640k_csr.json": "r04_lda_640k_csr_build.json",
         "bench_2ranks_one_card.json": "r04_final_bench_2ranks_one_card.json", "bench_2ranks_one_card_default.json": "r04_final_bench_2ranks_one_card_default.json"}
for c in ("cfg2", "cfg4", "cfg5", "lda640k"):
    pairs["ks_%s/%s_kernel_stats.csv" % (c, c)] = "r04_final_%s_kernel_stats.csv" % c
for f in glob.glob(os.path.join(E, "pmc_*.txt")):
    pairs[os.path.basename(f)] = "r04_" + os.path.basename(f)
for f in glob.glob(os.path.join(E, "traffic_*.json")):
    pairs[os.path.basename(f)] = "r04_" + os.path.basename(f)
for src, dst in sorted(pairs.items()):
    s = os.path.join(E, src)
    if not os.path.exists(s):
        hits = glob.glob(os.path.join(E, os.path.dirname(src), "**", os.path.basename(src)), recursive=True)
        s = hits[0] if hits else s
    if os.path.exists(s) and os.path.getsize(s) > 0:
        shutil.copyfile(s, os.path.join(P, dst)); print("%-44s -> profiles/%s" % (src, dst))
    else:
        print("MISSING or empty: %s" % src, file=sys.stderr)
