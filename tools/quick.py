import importlib, os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
P = importlib.import_module("fast-3d-pointcloud-segmentation_amd")
from conftest import CpuChecker
pts = P.synth_frame(0, 7, 160, 120, 30); prm = P.launch_params(voxel_res=0.02, seed_res=0.2)
ctx = P.Context(0)
print("ctx ok", flush=True)
lab = ctx.segment(pts, prm)
print("segment ok", ctx.result.as_dict(), flush=True)
ora = CpuChecker(os.path.join(ROOT, "oracle", "libf3ds_oracle.so"), "f3ds_oracle")
rc, olab, ores, oh = ora.segment(pts, prm)
print("equal", np.array_equal(lab, olab))
