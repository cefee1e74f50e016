#!/usr/bin/env python3
"""Randomized parity sweep on the CPU: the sequential emulation of the device algorithms (tests/emul) against the
literal oracle, all intermediate arrays.  usage: tools/fuzz_cpu.py [n_cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import conftest
P = conftest.pkg()
orc = conftest.CpuChecker(os.path.join(ROOT, "oracle", "libf3ds_oracle.so"), "f3ds_oracle")
em = conftest.CpuChecker(os.path.join(ROOT, "tests", "emul", "libf3ds_emul.so"), "f3ds_emul")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for it in range(n):
    w, hgt = int(rng.integers(20, 260)), int(rng.integers(20, 200))
    kind = int(rng.integers(0, 2))
    seed = int(rng.integers(1, 10**6))
    nan = int(rng.integers(0, 400)) if kind == 0 else 0
    vres = float(rng.choice([0.008, 0.01, 0.015, 0.02, 0.03, 0.05, 0.08]))
    kw = dict(voxel_res=vres, seed_res=vres * float(rng.choice([2, 3, 5, 8, 12, 20])), use_transform=int(rng.integers(0, 2)) if kind == 0 else 0,
              color_metric=int(rng.integers(0, 2)), geom_metric=int(rng.integers(0, 2)), merging=int(rng.integers(0, 3)), lambda_=float(rng.uniform(0.0, 1.0)),
              bins=int(rng.choice([0, 5, 20, 100, 500])), threshold=float(rng.choice([0.0, 0.05, 0.1, 0.3, 0.6, 1.0])), leaf_order=int(rng.integers(0, 2)),
              w_color=float(rng.uniform(0.0, 1.0)), w_spatial=float(rng.uniform(0.0, 1.0)), w_normal=float(rng.uniform(0.0, 6.0)))
    pts = P.synth_frame(kind, seed, w, hgt, nan)
    prm = P.launch_params(**kw)
    rc, olab, ores, oh = orc.segment(pts, prm)
    rc2, elab, eres, eh = em.segment(pts, prm)
    ok = rc == rc2
    if ok and rc == 0:
        ok = np.array_equal(olab, elab) and all(conftest.same_bits(oh.get(k), eh.get(k)) for k in conftest.ALL_DEBUG)
    if not ok:
        bad += 1
        print("MISMATCH case", it, "rc", rc, rc2, dict(kind=kind, seed=seed, w=w, h=hgt, nan=nan), kw, flush=True)
print("cases", n, "mismatches", bad)
