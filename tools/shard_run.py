#!/usr/bin/env python3
"""One configuration at the shard size of an N-GPU strong run, a fixed number of passes (for rocprofv3 --kernel-trace --stats / --pmc).
usage: python3 tools/shard_run.py CONFIG N [PASSES] ['{"solve_lanes": 8, "disable": ["ctm_fused_gauss"]}']   (the last: mmm_tuning_opts as JSON)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    c, n = int(sys.argv[1]), int(sys.argv[2])
    passes = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    import json
    tune = json.loads(sys.argv[4]) if len(sys.argv) > 4 else {}
    env = bench.Env(1)
    if tune:
        env.ctx.set_tuning(**tune)
    cfg = bench.CONFIGS[c]
    corpus = bench.make_corpus(c, cfg["docs"], 20261003 + (1 if c == 2 else c))
    r = bench.run_config(env, c, "weak", passes, 3, 3, 0, False, probe=False, proxy_shard=n, corpus=corpus)
    print("cfg %d shard 1/%d: %d documents, %.4f ms per step, kernels %s" % (c, n, r["config"]["docs_rank0"], r["ms_per_step"], r["iteration"]["kernel_us"]))
    env.close()


if __name__ == "__main__":
    main()
