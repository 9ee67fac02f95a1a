#!/usr/bin/env python3
"""One GPU: BASELINE's configurations at the shard sizes an N-GPU strong run gives every rank (D, D/2, D/4, D/8 documents) -- bench.py's
`also.shard_proxy`, one JSON line per (configuration, N) -> profiles/rNN_shard_sizes.jsonl.
usage: python3 tools/shard_sizes.py [--configs 2,4,5] [--shards 2,4,8] > gpurun_out/shard_sizes.jsonl"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="2,4,5")
    ap.add_argument("--shards", default="2,4,8")
    a = ap.parse_args()
    cfgs = [int(x) for x in a.configs.split(",")]
    shards = tuple(int(x) for x in a.shards.split(","))
    env = bench.Env(1)
    full = {}
    for c in cfgs:
        lda = bench.CONFIGS[c]["model"] == "lda"
        full[c] = bench.run_config(env, c, "weak", 20 if lda else 10, 5 if lda else 2, 5 if lda else 3, 0, False, probe=False)
    res = bench.shard_proxy(env, full, shards=shards, cfgs=cfgs)
    for k, rows in res.items():
        if k == "what":
            continue
        for r in rows:
            print(json.dumps(dict(r, config=k)))
    env.close()


if __name__ == "__main__":
    main()
