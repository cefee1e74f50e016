#!/usr/bin/env python3
"""BASELINE config 4 (the 20M-point fused scene, seed 3000, -v 0.02 -s 0.2 --NT --AL --CVX -t 0.2) segmented a few times on one context:
the program of the config-4 kernel-trace and PMC passes (profiles/r3_config4_*), and its per-stage device times.
usage: tools/config4_frame.py [repeats]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = importlib.import_module("fast-3d-pointcloud-segmentation_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
pts = P.synth_frame(1, 3000, 5000, 4000, 0)
prm = P.launch_params(voxel_res=0.02, seed_res=0.2, use_transform=0)
ctx = P.Context(0)
out = np.empty(len(pts), np.uint32)      # (the caller's label buffer, reused from call to call)
for i in range(n):
    t = time.perf_counter(); ctx.segment(pts, prm, labels_out=out); ms = (time.perf_counter() - t) * 1e3
    r = ctx.result
    print("scene %.2f ms host clock (%.1f Mpoints/s); device stages (voxelise, neighbours + normals, seeds, sweeps, supervoxels + adjacency, cluster + merge, labels): %s; normals kernel %.2f" % (
        ms, len(pts) / ms / 1e3, " ".join("%.2f" % x for x in list(r.ms_stage)[:7]), r.ms_stage[7]), r.as_dict() if i == 0 else "", flush=True)
