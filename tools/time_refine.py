#!/usr/bin/env python3
"""refineSupervoxels(3) after a frame: wall time on the GPU (row N3)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = importlib.import_module("fast-3d-pointcloud-segmentation_amd")
ctx = P.Context(0)
for name, pts, prm in (("1M frame", P.synth_frame(0, 1000, 1000, 1000, 30), P.launch_params()),
                       ("640x480", P.synth_frame(0, 2000, 640, 480, 200), P.launch_params()),
                       ("20M scene", P.synth_frame(1, 3000, 5000, 4000, 0), P.launch_params(voxel_res=0.02, seed_res=0.2, use_transform=0))):
    ctx.segment(pts, prm)
    for rep in range(3):
        t0 = time.perf_counter(); ctx.lib.f3ds_refine_supervoxels(ctx.handle, 3); dt = time.perf_counter() - t0
    r = ctx.refine_supervoxels(3)
    print("%s: V %d, supervoxels %d -> %d refined, refineSupervoxels(3) %.1f ms (segment %.1f ms)" % (name, ctx.result.n_voxels, ctx.result.n_supervoxels, len(r["label"]), dt * 1e3, ctx.result.ms_total), flush=True)
