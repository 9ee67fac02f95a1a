#!/usr/bin/env python3
"""Copies the summaries tools/refresh_evidence.sh left under gpurun_out/evidence/ into profiles/ under the round's names (MMM_ROUND, default r05)."""
import glob, os, shutil, sys
R_ = os.environ.get("MMM_ROUND", "r05")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E, P = os.path.join(ROOT, "gpurun_out", "evidence"), os.path.join(ROOT, "profiles")
pairs = {"bench_default.json": R_ + "_final_bench_default.json", "bench_cfg4.json": R_ + "_final_bench_cfg4.json", "bench_cfg5.json": R_ + "_final_bench_cfg5.json",
         "bench_cfg2_steps20.json": R_ + "_final_bench_cfg2_steps20.json", "lda_scaling.jsonl": R_ + "_lda_scaling.jsonl",
         "lda_640k_csr.json": R_ + "_lda_640k_csr_build.json",
         "bench_2ranks_one_card.json": R_ + "_final_bench_2ranks_one_card.json", "bench_2ranks_one_card_default.json": R_ + "_final_bench_2ranks_one_card_default.json"}
pairs["shard_sizes.jsonl"] = R_ + "_shard_sizes.jsonl"
pairs["solve_layouts.jsonl"] = R_ + "_solve_layouts.jsonl"
for c in ("cfg2", "cfg4", "cfg5", "lda640k"):
    pairs["ks_%s/%s_kernel_stats.csv" % (c, c)] = R_ + "_final_%s_kernel_stats.csv" % c
for f in glob.glob(os.path.join(E, "pmc_*.txt")):
    pairs[os.path.basename(f)] = R_ + "_" + os.path.basename(f)
for f in glob.glob(os.path.join(E, "traffic_*.json")):
    pairs[os.path.basename(f)] = R_ + "_" + os.path.basename(f)
for src, dst in sorted(pairs.items()):
    s = os.path.join(E, src)
    if not os.path.exists(s):
        hits = glob.glob(os.path.join(E, os.path.dirname(src), "**", os.path.basename(src)), recursive=True)
        s = hits[0] if hits else s
    if os.path.exists(s) and os.path.getsize(s) > 0:
        shutil.copyfile(s, os.path.join(P, dst)); print("%-44s -> profiles/%s" % (src, dst))
    else:
        print("MISSING or empty: %s" % src, file=sys.stderr)
