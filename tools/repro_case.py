#!/usr/bin/env python3
"""Run one frame on the GPU and on the oracle and name the arrays that differ (development aid).
usage: tools/repro_case.py kind seed width height nan_permille key=value ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, conftest
P = conftest.pkg()
kind, seed, w, h, nan = (int(x) for x in sys.argv[1:6])
kw = {}
for a in sys.argv[6:]:
    k, v = a.split("="); kw[k] = float(v) if "." in v or "e" in v else int(v)
pts = P.synth_frame(kind, seed, w, h, nan)
prm = P.launch_params(**kw)
orc = conftest.CpuChecker(os.path.join(ROOT, "oracle", "libf3ds_oracle.so"), "f3ds_oracle")
rc, olab, ores, oh = orc.segment(pts, prm)
ctx = P.Context(0)
try:
    lab = ctx.segment(pts, prm)
except Exception as ex:
    print("gpu error", ex, "oracle rc", rc); sys.exit(1)
print("oracle rc", rc, "labels equal", np.array_equal(lab, olab), {k: getattr(ctx.result, k) for k in ("n_voxels", "n_seeds", "n_supervoxels", "n_edges", "n_merges", "n_regions")},
      {k: getattr(ores, k) for k in ("n_supervoxels", "n_edges", "n_merges", "n_regions")})
for k in conftest.ALL_DEBUG:
    a, b = oh.get(k), ctx.debug(k)
    if not conftest.same_bits(a, b):
        n = min(len(a), len(b)); idx = np.nonzero(a[:n] != b[:n])[0]
        print("  differs:", k, len(a), len(b), "count", len(idx), "first", idx[:4], a[idx[:4]], b[idx[:4]])
