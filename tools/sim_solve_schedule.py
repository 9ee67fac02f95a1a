#!/usr/bin/env python3
"""What a schedule of the CTM solve phase built on the PREVIOUS pass' per-document evaluation counts would be worth: list-scheduling model of
k_ctm_solve_cpl's persistent waves (S document slots per wave in lock step, a slot refills when its solve stops; the nu phase of the wave's
range, then its lambda phase) on the counts dumped by tools/dump_nev.py.  Reports trips of the slowest wave (latency-bound reading) and of the
slowest SIMD (issue-bound reading: the OCC waves of a SIMD share its vector pipe) per schedule.
usage: python3 tools/sim_solve_schedule.py gpurun_out/r05/nev_cfg4.npz --slots 4 --occ 3 [--docs-div 8]"""
import argparse
import heapq

import numpy as np


def makespan(jobs, S):
    """trips of a wave whose S slots take the jobs in the given order"""
    if len(jobs) == 0:
        return 0
    h = [0] * S
    for j in jobs:
        t = heapq.heappop(h)
        heapq.heappush(h, t + int(j))
    return max(h)


def ranges_equal(D, W):
    base, rem = divmod(D, W)
    r0 = [w * base + min(w, rem) for w in range(W + 1)]
    return r0


def ranges_by_weight(wt, W):
    cs = np.concatenate([[0], np.cumsum(wt)])
    tgt = cs[-1] * np.arange(W + 1) / W
    return np.searchsorted(cs, tgt, side="left").clip(0, len(wt)).tolist()


def run(nu, lam, pnu, plam, S, W, occ, cost_nu, cost_lam, balance, lpt, chunk=64):
    D = len(nu)
    r = ranges_by_weight(pnu * cost_nu + plam * cost_lam, W) if balance else ranges_equal(D, W)
    r[0], r[-1] = 0, D
    tw = np.zeros(W)
    for w in range(W):
        a, b = r[w], r[w + 1]
        for true, pred, c in ((nu, pnu, cost_nu), (lam, plam, cost_lam)):
            jobs = true[a:b]
            if lpt:
                # the wave orders its range in chunks of <= 64 documents (one per lane) by predicted count, longest first
                idx = []
                for s in range(a, b, chunk):
                    e = min(b, s + chunk)
                    idx.extend((s + np.argsort(-pred[s:e], kind="stable")).tolist())
                jobs = true[idx]
            tw[w] += makespan(jobs, S) * c
    # SIMD = occ waves (blocks b, b + ncu, ...: wave i of each)
    nsimd = W // occ
    ts = tw[: nsimd * occ].reshape(occ, nsimd).sum(axis=0)
    return tw.max(), ts.max(), tw.mean(), (true_sum(nu, lam, cost_nu, cost_lam) / (S * W))


def true_sum(nu, lam, cn, cl):
    return float(nu.sum()) * cn + float(lam.sum()) * cl


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("npz")
    ap.add_argument("--slots", type=int, required=True)
    ap.add_argument("--occ", type=int, required=True)
    ap.add_argument("--cus", type=int, default=256)
    ap.add_argument("--docs-div", type=int, default=1, help="take the first D / n documents (the shard of an n-GPU strong run)")
    ap.add_argument("--cost", default="1,1", help="relative cost of a nu trip and a lambda trip")
    ap.add_argument("--pairs", default="3:4,8:9,30:31")
    a = ap.parse_args()
    z = np.load(a.npz)
    cn, cl = [float(x) for x in a.cost.split(",")]
    W = a.cus * 4 * a.occ
    for pr in a.pairs.split(","):
        p0, p1 = [int(x) for x in pr.split(":")]
        nu, lam, pnu, plam = [z[k].astype(np.int64) for k in ("nu_%d" % p1, "lam_%d" % p1, "nu_%d" % p0, "lam_%d" % p0)]
        D = len(nu) // a.docs_div
        nu, lam, pnu, plam = nu[:D], lam[:D], pnu[:D], plam[:D]
        print("pass %d predicted by pass %d: %d documents, %d waves x %d slots; corr nu %.2f lambda %.2f; ideal (sum / slots) %.1f trips" % (
            p1, p0, D, W, a.slots, np.corrcoef(nu, pnu)[0, 1], np.corrcoef(lam, plam)[0, 1], true_sum(nu, lam, cn, cl) / (a.slots * W)))
        for name, bal, lpt, pn, pl in (("as shipped (equal ranges, arrival order)", False, False, pnu, plam),
                                       ("longest-first inside the wave, predicted", False, True, pnu, plam),
                                       ("ranges balanced by predicted work", True, False, pnu, plam),
                                       ("both, predicted", True, True, pnu, plam),
                                       ("both, true counts (bound)", True, True, nu, lam)):
            mw, ms, mean, ideal = run(nu, lam, pn, pl, a.slots, W, a.occ, cn, cl, bal, lpt)
            print("  %-44s slowest wave %6.1f  slowest SIMD / occ %6.1f  mean wave %6.1f" % (name, mw, ms / a.occ, mean))




def pooled(nu, lam, S, nblocks, wpb):
    """blocks of wpb waves draw the documents of the block's contiguous range from one pool (phase by phase, a barrier between the phases):
    trips per block = the list-scheduling makespan over wpb * S slots (every wave of the block runs to about that)"""
    D = len(nu)
    r = ranges_equal(D, nblocks)
    t = np.zeros(nblocks)
    for b in range(nblocks):
        t[b] = makespan(nu[r[b]:r[b + 1]], S * wpb) + makespan(lam[r[b]:r[b + 1]], S * wpb)
    return t


def main_pooled():
    ap = argparse.ArgumentParser()
    ap.add_argument("npz"); ap.add_argument("--slots", type=int, required=True); ap.add_argument("--occ", type=int, required=True)
    ap.add_argument("--cus", type=int, default=256); ap.add_argument("--pairs", default="4,9,31"); ap.add_argument("--pooled", action="store_true")
    ap.add_argument("--docs-div", type=int, default=1)
    a = ap.parse_args()
    z = np.load(a.npz)
    for p in [int(x) for x in a.pairs.split(",")]:
        nu, lam = z["nu_%d" % p].astype(np.int64), z["lam_%d" % p].astype(np.int64)
        D = len(nu) // a.docs_div
        nu, lam = nu[:D], lam[:D]
        ideal = (nu.sum() + lam.sum()) / (a.slots * a.cus * 4 * a.occ)
        for name, nb, wpb in (("per wave (as shipped)", a.cus * 4 * a.occ, 1), ("pool per block of 4 waves", a.cus * a.occ, 4), ("pool per CU (one block of 4 occ waves)", a.cus, 4 * a.occ)):
            t = pooled(nu, lam, a.slots, nb, wpb)
            # SIMD total: occ waves; per wave / per 4-wave block: blocks b, b + cus, ... share a CU; per CU: every SIMD has occ waves of the block
            if wpb == 1:
                simd = t.reshape(a.occ, -1).sum(axis=0)
            elif wpb == 4:
                simd = t.reshape(a.occ, -1).sum(axis=0)
            else:
                simd = t * a.occ
            print("pass %2d  %-40s ideal %6.1f  mean trips %6.1f  slowest unit %6.1f  slowest SIMD / occ %6.1f" % (p, name, ideal, t.mean(), t.max(), simd.max() / a.occ))


if __name__ == "__main__":
    import sys
    if "--pooled" in sys.argv:
        main_pooled()
    else:
        main()
