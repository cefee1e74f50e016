#!/usr/bin/env python3
"""Merge stage of one 1M-point frame (and of a batch) with every merge kernel layout: device ms of the stage.
With F3DS_LIB pointing at a `make PROF=1` build the kernels also print their per-phase shader-clock totals."""
import os as _os; _os.environ.setdefault("F3DS_DEV", "1")      # this tool drives development switches (csrc/f3ds_dev.h)
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
P = importlib.import_module("fast-3d-pointcloud-segmentation_amd")
prm = P.launch_params(voxel_res=0.008, seed_res=0.08)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
frames = [P.synth_frame(0, 1000 + i, 1000, 1000, 30) for i in range(min(nb, 8))]
ctxs = [P.Context(0) for _ in range(nb)]
VARIANTS = [dict(F3DS_MERGE_NW=nw, F3DS_MERGE_KEYS=k) for nw in ("8", "4") for k in ("lds", "global")] + [dict(F3DS_FORCE_GLOBAL_MERGE="1")]
for v in VARIANTS:
    for k in ("F3DS_FORCE_GLOBAL_MERGE", "F3DS_MERGE_NW", "F3DS_MERGE_KEYS"):
        os.environ.pop(k, None)
    os.environ.update(v)
    for rep in range(2):
        t0 = time.perf_counter()
        P.segment_batch(ctxs, [frames[i % len(frames)] for i in range(nb)], prm)
        dt = (time.perf_counter() - t0) * 1e3
    r = ctxs[0].result
    print("%-60s frames %3d  merge stage %8.2f ms   whole call %8.2f ms  (merges %d)" % (v, nb, r.ms_stage[5], dt, r.n_merges), flush=True)
