"""(GPU box) stage 0's tile path on the 64 bench frames: most distinct voxels in a tile, descriptors, leaves (printed by the library under F3DS_TRACE_ERR=1)."""
import os as _os; _os.environ.setdefault("F3DS_DEV", "1")      # this tool drives development switches (csrc/f3ds_dev.h)
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["F3DS_TRACE_ERR"] = "1"
P = importlib.import_module("fast-3d-pointcloud-segmentation_amd")
prm = P.launch_params(voxel_res=0.008, seed_res=0.08)
ctx = P.Context(0)
for seed in range(1000, 1000 + int(sys.argv[1]) if len(sys.argv) > 1 else 1064):
    ctx.segment(P.synth_frame(0, seed, 1000, 1000, 30), prm)
