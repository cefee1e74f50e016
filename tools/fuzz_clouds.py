#!/usr/bin/env python3
"""Randomized parity on hand-made degenerate clouds (empty, one point, duplicates, lines, planes, NaN / inf, negative z,
huge extents): device algorithms (CPU emulation by default, the GPU with --gpu) against the oracle.
usage: tools/fuzz_clouds.py [cases] [seed] [--gpu]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, conftest
P = conftest.pkg()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
gpu = "--gpu" in sys.argv
n_cases = int(args[0]) if args else 200
rng = np.random.default_rng(int(args[1]) if len(args) > 1 else 1)
orc = conftest.CpuChecker(os.path.join(ROOT, "oracle", "libf3ds_oracle.so"), "f3ds_oracle")
em = None if gpu else conftest.CpuChecker(os.path.join(ROOT, "tests", "emul", "libf3ds_emul.so"), "f3ds_emul")
ctx = P.Context(0) if gpu else None


def cloud(rng):
    kind = int(rng.integers(0, 8))
    n = int(rng.choice([0, 1, 2, 3, 7, 50, 400, 3000]))
    s = float(rng.choice([0.01, 0.1, 1.0, 4.0]))
    xyz = rng.uniform(-s, s, (n, 3)).astype(np.float32)
    xyz[:, 2] = np.abs(xyz[:, 2]) + np.float32(rng.choice([0.0, 0.5, 2.0]))
    if kind == 1 and n: xyz[:] = xyz[0]                                   # all points identical
    if kind == 2 and n: xyz[:, 1] = xyz[0, 1]; xyz[:, 2] = xyz[0, 2]      # a line
    if kind == 3 and n: xyz[:, 2] = xyz[0, 2]                             # a plane
    if kind == 4 and n: xyz = xyz[rng.integers(0, max(1, n // 10), n)]    # many duplicates
    if kind == 5 and n: xyz[rng.random(n) < 0.3] = np.nan                 # NaNs
    if kind == 6 and n: xyz[:, 2] *= -1                                   # negative z (main() folds it)
    if kind == 7 and n: xyz[rng.integers(0, n)] = [np.inf, 0, 1]          # an infinite coordinate
    rgba = rng.integers(0, 1 << 24, n).astype(np.uint32)
    if int(rng.integers(0, 3)) == 0 and n: rgba[:] = rgba[0]
    out = np.zeros((n, 4), np.float32); out[:, :3] = xyz; out[:, 3] = rgba.view(np.float32)
    return out, s


bad = 0; n_ok = 0; n_nonempty = 0
for it in range(n_cases):
    pts, s = cloud(rng)
    vres = s * float(rng.choice([0.02, 0.05, 0.2, 1.0]))
    kw = dict(voxel_res=vres, seed_res=vres * float(rng.choice([1, 2, 3, 8])), use_transform=int(rng.integers(0, 2)), color_metric=int(rng.integers(0, 2)),
              geom_metric=int(rng.integers(0, 2)), merging=int(rng.integers(0, 3)), lambda_=float(rng.uniform(0, 1)), bins=int(rng.choice([0, 5, 100])),
              threshold=float(rng.choice([0.0, 0.2, 1.0])), leaf_order=int(rng.integers(0, 2)))
    prm = P.launch_params(**kw)
    rc, olab, ores, oh = orc.segment(pts, prm)
    if gpu:
        rc2 = 0
        try:
            elab = ctx.segment(pts, prm)
            get = ctx.debug
        except Exception as ex:
            rc2 = getattr(ex, "code", -99); elab = None
    else:
        rc2, elab, eres, eh = em.segment(pts, prm)
        get = eh.get
    ok = rc == rc2
    n_ok += rc == 0; n_nonempty += rc == 0 and ores.n_supervoxels > 1
    if ok and rc == 0:
        ok = np.array_equal(olab, elab) and (len(pts) == 0 or ores.n_voxels == 0 or all(conftest.same_bits(oh.get(k), get(k)) for k in conftest.ALL_DEBUG))
    if not ok:
        bad += 1
        print("MISMATCH case", it, "rc", rc, rc2, "n", len(pts), kw, flush=True)
        np.save("/tmp/fuzz_cloud_%d.npy" % it, pts)
print("cases", n_cases, "accepted by the oracle", n_ok, "with more than one supervoxel", n_nonempty, "mismatches", bad)
