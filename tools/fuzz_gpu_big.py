"""Randomized parity on the GPU with mid-size frames (100k-350k points): every intermediate array against the oracle.
usage: tools/fuzz_gpu_big.py [cases] [seed] [min seed / voxel resolution]      (the third argument keeps to big supervoxels: the merge loop's wide speculative merges)
refineSupervoxels is compared too unless the frame has more than 20 000 seeds (the ORACLE's refine takes minutes there: run under `timeout`)."""
import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, conftest
P = conftest.pkg()
orc = conftest.CpuChecker(os.path.join(ROOT, "oracle", "libf3ds_oracle.so"), "f3ds_oracle")
ctx = P.Context(0)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 31337); bad=0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 14):
    w, hgt = int(rng.integers(300, 700)), int(rng.integers(250, 520))
    kind = int(rng.integers(0, 2)); seed=int(rng.integers(1, 10**6))
    vres = float(rng.choice([0.006, 0.008, 0.012, 0.02]))
    ratios = [r for r in (2, 3, 6, 10, 16, 24, 40, 60) if r >= (float(sys.argv[3]) if len(sys.argv) > 3 else 0)]
    kw = dict(voxel_res=vres, seed_res=vres * float(rng.choice(ratios)), use_transform=int(rng.integers(0, 2)) if kind == 0 else 0,
              color_metric=int(rng.integers(0, 2)), geom_metric=int(rng.integers(0, 2)), merging=int(rng.integers(0, 3)), lambda_=float(rng.uniform(0.0, 1.0)),
              bins=int(rng.choice([0, 50, 500])), threshold=float(rng.choice([0.1, 0.2, 0.5])), leaf_order=int(rng.integers(0, 2)))
    pts = P.synth_frame(kind, seed, w, hgt, int(rng.integers(0, 200)) if kind == 0 else 0)
    prm = P.launch_params(**kw)
    rc, olab, ores, oh = orc.segment(pts, prm); rc2 = 0
    try:
        elab = ctx.segment(pts, prm); eres = ctx.result
    except Exception as ex:
        rc2 = getattr(ex, "code", -99); elab = None
    ok = rc == rc2 and (rc != 0 or (np.array_equal(olab, elab) and all(conftest.same_bits(oh.get(k), ctx.debug(k)) for k in conftest.ALL_DEBUG)))
    if ok and rc == 0 and ores.n_seeds <= 20000:          # row N3 on the same frame
        k = int(rng.integers(1, 4))
        want = oh.refine(k); got = ctx.refine_supervoxels(k)
        ok = all(want[key].shape == got[key].shape and conftest.same_bits(want[key], got[key]) for key in want)
        if not ok: print("  refine(%d) differs" % k)
    print(it, "ok" if ok else "MISMATCH", rc, rc2, w, hgt, kind, seed, kw, "V", ores.n_voxels, "S", ores.n_seeds, "merges", ores.n_merges, flush=True)
    bad += not ok
print("bad", bad)
