"""(GPU box) how many 128-voxel tiles of the bench frames overflow the LDS tables of d_normals_t / the sweeps (F3DS_DBG_TILE_LIST_LEN = 0xFFFFFFFF): they take the global-memory paths."""
import os as _os; _os.environ.setdefault("F3DS_DEV", "1")
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("fast-3d-pointcloud-segmentation_amd")
prm = P.launch_params(voxel_res=0.008, seed_res=0.08)
ctx = P.Context(0)
for seed in (1000, 1007, 1033, 1063):
    ctx.segment(P.synth_frame(0, seed, 1000, 1000, 30), prm)
    t = ctx.tile_list_lengths()
    ok = t[t != 0xFFFFFFFF]
    print("seed", seed, "tiles", len(t), "overflowed", int((t == 0xFFFFFFFF).sum()), "one-ring list: mean %.0f max %d" % (ok.mean(), ok.max()))
