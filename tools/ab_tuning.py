#!/usr/bin/env python3
"""A/B of mmm_tuning_opts settings in ONE process on one GPU: the configurations alternate, `rounds` times each, on the same corpus shard.
usage: python3 tools/ab_tuning.py CONFIG N ROUNDS '{"side_stream": -1}' '{"side_stream": 0}' ...   (N: shard 1/N of the corpus; 1 = whole)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    c, n, rounds = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    settings = [json.loads(a) for a in sys.argv[4:]]
    env = bench.Env(1)
    cfg = bench.CONFIGS[c]
    lda = cfg["model"] == "lda"
    corpus = bench.make_corpus(c, cfg["docs"], 20261003 + (1 if c == 2 else c))
    res = {i: [] for i in range(len(settings))}
    for _ in range(rounds):
        for i, s in enumerate(settings):
            env.ctx.set_tuning(**s)
            r = bench.run_config(env, c, "weak", 20 if lda else 10, 5 if lda else 2, 5 if lda else 3, 0, False, probe=False, proxy_shard=n, corpus=corpus)
            res[i].append(r["ms_per_step"])
    for i, s in enumerate(settings):
        v = sorted(res[i])
        print(json.dumps({"config": c, "shard": "1/%d" % n, "tuning": s, "ms_per_step_median": v[len(v) // 2], "ms_per_step_min": v[0], "all": res[i]}))
    env.close()


if __name__ == "__main__":
    main()
