#!/usr/bin/env python3
"""One BASELINE config-2 frame (seed 1000, 1M points, -v 0.008 -s 0.08 --AL --CVX -t 0.2) segmented a few times on one context: the program
of the single-kernel PMC passes, and the per-stage device times of a lone frame.  usage: tools/lone_frame.py [repeats]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = importlib.import_module("fast-3d-pointcloud-segmentation_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
pts = P.synth_frame(0, 1000, 1000, 1000, 30)
prm = P.launch_params(voxel_res=0.008, seed_res=0.08)
ctx = P.Context(0)
for i in range(n):
    t = time.perf_counter(); ctx.segment(pts, prm); ms = (time.perf_counter() - t) * 1e3
    r = ctx.result
    print("frame %.2f ms host clock; device stages (voxelise, neighbours + normals, seeds, sweeps, supervoxels + adjacency, cluster + merge, labels): %s" % (ms, " ".join("%.2f" % x for x in list(r.ms_stage)[:7])), r.as_dict() if i == 0 else "")
