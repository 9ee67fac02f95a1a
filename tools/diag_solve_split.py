#!/usr/bin/env python3
"""How the solve phase of a CTM pass splits between update_nu! and update_lambda! (the two LD_MMA solves, MMCTM.jl:127-170), per build:
after P warm-up passes the stage calls mmm_ctm_update_zeta / theta / nu / lambda are run once with HIP events around the solve launches.
usage: python tools/diag_solve_split.py [config 4|5] [docs] [passes]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mmm_pkg, np_ref
import bench

cfg_id = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = bench.CONFIGS[cfg_id]
D = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["docs"]
P = int(sys.argv[3]) if len(sys.argv) > 3 else 20
pkg = mmm_pkg.load(); lib = pkg.lib(); chk = pkg._lib.check
ctx = pkg.Context(0)
K, V = cfg["K"], cfg["V"]
X, g0 = np_ref.synth_mm(D, V, K, seed=20261003 + cfg_id)
if cfg["model"] == "mmctm":
    m = pkg.MMCTM(K, [0.1] * len(K), V, X, γ0=g0, ctx=ctx)
else:
    GM = sum(K[i] * int(f.max(axis=0).sum()) for i, f in enumerate(bench.snv3()))
    m = pkg.IMMCTM(K, [0.1], bench.snv3(), X, γ0=np.random.default_rng(1).integers(1, 101, size=GM).astype(np.float64), ctx=ctx)
chk(lib.mmm_ctm_iterate(m._h, P, 1), ctx.h)
out = {"config": cfg_id, "docs": D, "passes": P, "geometry": m.geometry(), "env": {k: v for k, v in os.environ.items() if k.startswith("MMM_")}}
ctx.profile_begin(); chk(lib.mmm_ctm_iterate(m._h, 4, 1), ctx.h); n, ms = ctx.profile_end()
out["fused_solve_us"] = ms / n * 1e3
st = m.solver_stats()
out["evals_per_doc"] = {"nu": st["n_eval_nu"] / D, "lambda": st["n_eval_lambda"] / D}
chk(lib.mmm_ctm_update_zeta(m._h), ctx.h); chk(lib.mmm_ctm_update_theta(m._h), ctx.h)
ctx.profile_begin(); chk(lib.mmm_ctm_update_nu(m._h), ctx.h); n, ms = ctx.profile_end(); out["nu_us"] = ms * 1e3; out["nu_launches"] = n
ctx.profile_begin(); chk(lib.mmm_ctm_update_lambda(m._h), ctx.h); n, ms = ctx.profile_end(); out["lambda_us"] = ms * 1e3; out["lambda_launches"] = n
st = m.solver_stats()
out["stage_evals_per_doc"] = {"nu": st["n_eval_nu"] / D, "lambda": st["n_eval_lambda"] / D}
print(json.dumps(out))
