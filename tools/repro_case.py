import os, sys
ROOT='/root/repo'
ROOT=os.environ.get("GRAFT_REPO_ROOT", ROOT); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, conftest
P = conftest.pkg()
kw={'voxel_res': 0.012, 'seed_res': 0.036000000000000004, 'use_transform': 0, 'color_metric': 1, 'geom_metric': 1, 'merging': 0, 'lambda_': 0.40850214837200993, 'bins': 0, 'threshold': 0.1, 'leaf_order': 0}
pts = P.synth_frame(0, 850543, 551, 346, int(sys.argv[1]) if len(sys.argv)>1 else 0)
ctx = P.Context(0)
try:
    ctx.segment(pts, P.launch_params(**kw)); print("ok", ctx.result.as_dict())
except Exception as ex:
    print("error", ex, ctx.result.as_dict())
