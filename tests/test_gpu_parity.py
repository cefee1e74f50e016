"""GPU parity proper: libf3ds (HIP, through the C-ABI) against the CPU oracle, bit for bit, on every
intermediate array the two expose.  Sizes are chosen so the oracle finishes in seconds."""
import numpy as np
import pytest

from conftest import ALL_DEBUG, FIXTURE_PCD, first_mismatch

pytestmark = pytest.mark.gpu

CASES = {
    # name: (synth args or 'fixture', param overrides)
    "rgbd_160x120": ((0, 7, 160, 120, 30), dict(voxel_res=0.02, seed_res=0.2)),
    "rgbd_320x240_ghosts": ((0, 11, 320, 240, 50), dict(voxel_res=0.012, seed_res=0.1)),
    "rgbd_160x120_desc_leaf_order": ((0, 7, 160, 120, 30), dict(voxel_res=0.02, seed_res=0.2, leaf_order=1)),
    "rgbd_160x120_no_transform": ((0, 7, 160, 120, 30), dict(voxel_res=0.02, seed_res=0.2, use_transform=0)),
    "rgbd_160x120_rgb_metric": ((0, 7, 160, 120, 30), dict(voxel_res=0.02, seed_res=0.2, color_metric=1)),
    "rgbd_160x120_equalization": ((0, 7, 160, 120, 30), dict(voxel_res=0.02, seed_res=0.2, merging=2)),
    "rgbd_160x120_manual_lambda": ((0, 7, 160, 120, 30), dict(voxel_res=0.02, seed_res=0.2, merging=0, lambda_=0.3)),
    "fused_200k_nan_lambda": ((1, 3000, 400, 500, 0), dict(voxel_res=0.04, seed_res=0.4, use_transform=0)),
    "fixture_launch_flags": ("fixture", {}),
}


def _points(P, spec):
    if spec == "fixture":
        return P.read_pcd(FIXTURE_PCD)
    return P.synth_frame(*spec)


@pytest.mark.parametrize("name", list(CASES))
def test_every_stage_matches_oracle(P, oracle, gpu_ctx, name):
    spec, kw = CASES[name]
    pts = _points(P, spec)
    prm = P.launch_params(**kw)
    rc, olab, ores, oh = oracle.segment(pts, prm)
    assert rc == 0
    glab = gpu_ctx.segment(pts, prm)
    gres = gpu_ctx.result
    for f in ("n_points", "n_finite", "n_voxels", "octree_depth", "n_seed_cells", "n_seeds", "n_supervoxels", "n_edges", "n_merges", "n_regions", "sweeps"):
        assert getattr(gres, f) == getattr(ores, f), f
    problems = []
    for what in ALL_DEBUG:
        m = first_mismatch(what, oh.get(what), gpu_ctx.debug(what))
        if m:
            problems.append(m)
    assert not problems, "\n".join(problems)
    assert np.array_equal(olab, glab)
    assert (np.isnan(ores.lambda_) and np.isnan(gres.lambda_)) or ores.lambda_ == gres.lambda_
    ox, ol, oc = oh.voxel_cloud()
    gx, gl, gc = gpu_ctx.voxel_cloud()
    assert np.array_equal(ox.view(np.uint32), gx.view(np.uint32)) and np.array_equal(ol, gl) and np.array_equal(oc, gc)
