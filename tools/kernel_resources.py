#!/usr/bin/env python3
"""Register / scratch / LDS figures of the shipped kernels, from `hipcc -Rpass-analysis=kernel-resource-usage` (no GPU needed).
usage: python tools/kernel_resources.py [substring ...]   -> one line per kernel whose (demangled) name contains a substring"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "multimodalmusig.jl_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=off -Wno-unused-function -Wno-pass-failed".split()
want = sys.argv[1:]
for src in ("lda.hip", "ctm.hip", "p2p.hip"):
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-S", "--cuda-device-only", "-o", os.path.join(td, "x.s"), os.path.join(CS, src),
                            "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    for blk in r.stderr.split("remark: Function Name: ")[1:]:
        name = blk.split()[0]
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
        dn = re.sub(r"\(.*$", "", dn).replace("void ", "")
        if want and not any(w in dn for w in want):
            continue
        g = lambda k: re.search(k + r": (\d+)", blk).group(1)
        print("%-64s VGPRs %3s  spilled %3s  scratch %4s B/lane  waves/SIMD %s  static LDS %6s B" % (
            dn[:64], g("VGPRs"), g("VGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
