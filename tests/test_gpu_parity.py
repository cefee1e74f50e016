"""GPU parity proper: libf3ds (HIP, through the C-ABI) against the CPU oracle and the committed
golden file, bit for bit, on every intermediate array.  Sizes: the oracle finishes in seconds."""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import sha_of, ALL_DEBUG, ROOT, first_mismatch, same_bits
from golden_cases import GOLDEN_CASES, case_params, case_points

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_golden.json")))


@pytest.mark.parametrize("stage0", ["sort", "tiles"])
@pytest.mark.parametrize("name", list(GOLDEN_CASES))
def test_every_stage_matches_oracle_and_golden(P, oracle, gpu_ctx, name, stage0, monkeypatch):
    """Every intermediate array of every golden case, with stage 0 (voxelisation) on either of its paths: the three-pass radix sort of all points (what a lone frame
    takes) and the tile path (per-tile grouping in LDS + a sort of the tiles' voxel descriptors: what the frames of a batch take; F3DS_VOX_TILES=2 forces it here)."""
    monkeypatch.setenv("F3DS_VOX_TILES", "2" if stage0 == "tiles" else "0")
    pts = case_points(P, name)
    prm = case_params(P, name)
    rc, olab, ores, oh = oracle.segment(pts, prm)
    assert rc == 0
    glab = gpu_ctx.segment(pts, prm)
    gres = gpu_ctx.result
    for f in ("n_points", "n_finite", "n_voxels", "octree_depth", "n_seed_cells", "n_seeds", "n_supervoxels", "n_edges", "n_merges", "n_regions", "sweeps"):
        assert getattr(gres, f) == getattr(ores, f), f
    problems = []
    for what in ALL_DEBUG:
        got = gpu_ctx.debug(what)
        m = first_mismatch(what, oh.get(what), got)
        if m:
            problems.append(m)
        if not m:       # equal to the oracle run here: then equal to the committed hash too (a mismatch is reported once, below)
            assert sha_of(got) == GOLD[name]["sha256"][what], what
    assert not problems, "\n".join(problems)
    assert np.array_equal(olab, glab)
    assert sha_of(glab) == GOLD[name]["labels_sha256"]
    assert (np.isnan(ores.lambda_) and np.isnan(gres.lambda_)) or ores.lambda_ == gres.lambda_
    ox, ol, oc = oh.voxel_cloud()
    gx, gl, gc = gpu_ctx.voxel_cloud()
    assert np.array_equal(ox.view(np.uint32), gx.view(np.uint32)) and np.array_equal(ol, gl) and np.array_equal(oc, gc)


def test_tile_voxelisation_falls_back_where_it_must_and_agrees_everywhere(P, oracle, monkeypatch):
    """Stage 0's tile path refuses, per context and for good, what it is not made for -- an unorganised cloud (no locality: a tile holds more distinct voxels than its
    LDS table), voxels of more than 256 points (the per-leaf order check would go quadratic) -- and the frame takes the sort path in the same call; either way the arrays
    are the oracle's.  Also: a 1000-wide frame with many holes (runs of one voxel meet inside one 256-point step: the leaf lists are sorted in place), both leaf orders,
    and a batch that mixes frames of both kinds."""
    monkeypatch.setenv("F3DS_VOX_TILES", "2")
    cases = [("organised, holes", P.synth_frame(0, 21, 1000, 120, 150), dict(voxel_res=0.008, seed_res=0.08)),
             ("organised, descending leaves", P.synth_frame(0, 22, 640, 200, 30), dict(voxel_res=0.01, seed_res=0.1, leaf_order=1)),
             ("organised, no transform", P.synth_frame(0, 23, 500, 300, 30), dict(voxel_res=0.02, seed_res=0.2, use_transform=0)),
             ("dense voxels", P.synth_frame(0, 24, 400, 300, 10), dict(voxel_res=0.08, seed_res=0.4)),
             ("unorganised cloud", P.synth_frame(1, 3001, 300, 400, 0), dict(voxel_res=0.04, seed_res=0.4, use_transform=0))]
    for what, pts, kw in cases:
        prm = P.launch_params(**kw)
        ctx = P.Context(0)
        for rep in range(2):      # (the second call of a context that fell back goes straight to the sort path)
            rc, olab, ores, oh = oracle.segment(pts, prm)
            assert rc == 0
            glab = ctx.segment(pts, prm)
            assert np.array_equal(olab, glab), what
            for w in ("GRID", "VOXEL_KEYS", "VOXEL_COUNT", "VOXEL_XYZ", "VOXEL_RGB", "POINT_VOXEL", "VOXEL_NEIGHBORS", "VOXEL_NORMAL", "MERGES"):
                assert first_mismatch(w, oh.get(w), ctx.debug(w)) is None, (what, w)
        ctx.close()
    frames = [c[1] for c in cases[:3]] + [cases[4][1], np.zeros((0, 4), np.float32), cases[0][1]]
    prm = P.launch_params(voxel_res=0.012, seed_res=0.12)
    ctxs = [P.Context(0) for _ in frames]
    got = P.segment_batch(ctxs, frames, prm)
    for f, g in zip(frames, got):
        rc, olab, _, _ = oracle.segment(f, prm)
        assert rc == 0 and np.array_equal(olab, g)
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("stage0", ["sort", "tiles"])
def test_deep_grids_take_the_two_word_voxel_table(P, oracle, monkeypatch, stage0):
    """The voxel table of the neighbour search keeps key and ordinal in one 8-byte word for grids of depth <= 10 (every BASELINE configuration) and falls back to 63-bit keys
    + a value array beyond: a frame voxelised at 0.5 mm (depth 12) and the same frame at 2 mm (depth 10) against the oracle, neighbour table included."""
    monkeypatch.setenv("F3DS_VOX_TILES", "2" if stage0 == "tiles" else "0")
    pts = P.synth_frame(0, 5, 200, 150, 20)
    ctx = P.Context(0)
    for vres, depth in ((0.0005, 12), (0.002, 10)):
        prm = P.launch_params(voxel_res=vres, seed_res=vres * 10)
        rc, olab, ores, oh = oracle.segment(pts, prm)
        assert rc == 0 and ores.octree_depth == depth
        glab = ctx.segment(pts, prm)
        assert ctx.result.octree_depth == depth and np.array_equal(olab, glab)
        for w in ("GRID", "VOXEL_KEYS", "VOXEL_COUNT", "VOXEL_NEIGHBORS", "VOXEL_NORMAL", "SEED_KEPT", "VOXEL_SVLABEL", "MERGES"):
            assert first_mismatch(w, oh.get(w), ctx.debug(w)) is None, (vres, w)
    ctx.close()


def test_global_memory_merge_kernel_matches_too(P, oracle, monkeypatch):
    """d_merge (edges in HBM, used when they do not fit LDS) against the oracle."""
    monkeypatch.setenv("F3DS_FORCE_GLOBAL_MERGE", "1")
    ctx = P.Context(0)
    for name in ("rgbd_320x240_ghosts", "fixture_launch_flags"):
        pts = case_points(P, name); prm = case_params(P, name)
        rc, olab, ores, oh = oracle.segment(pts, prm)
        glab = ctx.segment(pts, prm)
        assert np.array_equal(olab, glab)
        assert not first_mismatch("MERGES", oh.get("MERGES"), ctx.debug("MERGES"))
        ox, ol, _ = oh.voxel_cloud(); gx, gl, _ = ctx.voxel_cloud()
        assert np.array_equal(ox.view(np.uint32), gx.view(np.uint32)) and np.array_equal(ol, gl)
    ctx.close()


def test_point_sort_with_key_index_pairs_matches_too(P, oracle, monkeypatch):
    """The point sort packs (Morton code, point index) into one 64-bit word when both fit; the (key, index) pair path
    that takes over when they do not is forced here through F3DS_SORT_PAIRS (read per call)."""
    monkeypatch.setenv("F3DS_SORT_PAIRS", "1")
    ctx = P.Context(0)
    for name in ("rgbd_320x240_ghosts", "fixture_launch_flags", "rgbd_160x120_equalization"):
        pts = case_points(P, name); prm = case_params(P, name)
        rc, olab, ores, oh = oracle.segment(pts, prm)
        glab = ctx.segment(pts, prm)
        assert np.array_equal(olab, glab)
        for what in ("VOXEL_KEYS", "VOXEL_COUNT", "VOXEL_XYZ", "VOXEL_RGB", "POINT_VOXEL"):
            assert not first_mismatch(what, oh.get(what), ctx.debug(what))
    ctx.close()


def test_voxel_sums_as_two_kernels_match_too(P, oracle, monkeypatch):
    """The voxel sums gather their points themselves (d_voxel_gather_accum, segment table from d_seg_count / d_seg_write); the earlier chain
    (d_heads, scan, d_segstart, d_point_gather, d_voxel_accum) stays behind F3DS_SPLIT_VOXEL_ACCUM (read per call) for A/B runs: same arrays."""
    monkeypatch.setenv("F3DS_SPLIT_VOXEL_ACCUM", "1")
    ctx = P.Context(0)
    for name in ("rgbd_320x240_ghosts", "fixture_launch_flags", "rgbd_160x120_equalization"):
        pts = case_points(P, name); prm = case_params(P, name)
        rc, olab, ores, oh = oracle.segment(pts, prm)
        glab = ctx.segment(pts, prm)
        assert np.array_equal(olab, glab)
        for what in ("VOXEL_KEYS", "VOXEL_COUNT", "VOXEL_XYZ", "VOXEL_RGB", "POINT_VOXEL"):
            assert not first_mismatch(what, oh.get(what), ctx.debug(what))
    ctx.close()


def test_recluster_and_clustering_mirror(P, oracle, gpu_ctx):
    """Clustering::cluster(threshold) again on the same supervoxels, with other metrics."""
    pts = case_points(P, "rgbd_160x120")
    sv = P.SupervoxelClustering(0.02, 0.2, context=gpu_ctx)
    sv.setUseSingleCameraTransform(True); sv.setInputCloud(pts)
    sv.setColorImportance(0.2); sv.setSpatialImportance(0.4); sv.setNormalImportance(1.0)
    seg = P.Clustering()
    seg.set_delta_g(P.CONVEX_NORMALS_DIFF)
    seg.set_initialstate(sv)
    seg.cluster(0.2)
    rc, olab, ores, oh = oracle.segment(pts, case_params(P, "rgbd_160x120"))
    assert np.array_equal(seg.get_point_labels(), olab) and seg.get_lambda() == ores.lambda_
    for name in ("rgbd_160x120_rgb_metric", "rgbd_160x120_equalization", "rgbd_160x120_manual_lambda", "rgbd_160x120_threshold_1"):
        prm = case_params(P, name)
        seg.set_delta_c(prm.color_metric); seg.set_delta_g(prm.geom_metric); seg.set_merging(prm.merging)
        if prm.merging == P.MANUAL_LAMBDA and prm.lambda_:
            seg.set_lambda(prm.lambda_)
        seg.cluster(prm.threshold)                     # second call: recluster on the device-resident supervoxels
        rc, ol2, or2, oh2 = oracle.segment(pts, prm)
        assert np.array_equal(seg.get_point_labels(), ol2), name
        xyz, lab = seg.get_labeled_cloud()
        ox, ol, _ = oh2.voxel_cloud()
        assert np.array_equal(xyz.view(np.uint32), ox.view(np.uint32)) and np.array_equal(lab, ol), name


def test_edge_cases(P, oracle, gpu_ctx):
    prm = P.launch_params(voxel_res=0.02, seed_res=0.2)
    nan = np.float32("nan")
    cases = {
        "empty": np.zeros((0, 4), np.float32),
        "all_nan": np.full((10, 4), nan, np.float32),
        "single": np.array([[0.1, 0.2, 1.0, 0]], np.float32),
        "two_identical": np.array([[0.1, 0.2, 1.0, 0], [0.1, 0.2, 1.0, 0]], np.float32),
        "z_zero_and_inf": np.array([[0.1, 0.2, 0.0, 0], [0.3, 0.1, 1.0, 0], [0.3, 0.1, np.inf, 0], [0.5, 0.5, 2.0, 0]], np.float32),
        "negative_z_folded": np.array([[0.1, 0.2, -1.0, 0], [0.1, 0.2, 1.0, 0]], np.float32),
        "tiny_plane": P.synth_frame(0, 3, 24, 18, 0),
    }
    for name, pts in cases.items():
        rc, olab, ores, _ = oracle.segment(pts, prm)
        assert rc == 0, name
        glab = gpu_ctx.segment(pts, prm)
        assert np.array_equal(olab, glab), name
        assert gpu_ctx.result.n_voxels == ores.n_voxels and gpu_ctx.result.n_regions == ores.n_regions, name
    with pytest.raises(P.F3dsError):
        gpu_ctx.segment(cases["single"], P.launch_params(voxel_res=0.0))
    with pytest.raises(ValueError):
        gpu_ctx.segment(cases["tiny_plane"], P.launch_params(voxel_res=0.02, seed_res=0.2, merging=0, lambda_=1.5))


def test_points_on_cell_borders_get_the_reference_keys(P, oracle, emul, gpu_ctx):
    """Voxel keys are (unsigned)((x - min) / res) in double.  The key kernel multiplies by 1 / res wherever the product provably truncates like the quotient and divides
    only within 1e-6 of a cell border (csrc/f3ds_numerics.h, n_point_key): a lattice of points exactly ON the borders (and one ulp to either side) takes the division
    path on nearly every coordinate; voxel keys, counts and labels must still be the oracle's bits -- also for a resolution whose reciprocal is inexact, and at depth > 10
    (the 64-bit Morton interleave)."""
    rng = np.random.default_rng(5)
    for res, span in ((0.008, 40), (0.013, 30), (0.002, 1400)):
        r32 = np.float32(res)
        i = rng.integers(0, span, (6000, 3)).astype(np.float32)
        base = (i * r32).astype(np.float32)                                  # on (or within a rounding of) a border
        nudge = rng.integers(-1, 2, base.shape)
        pts3 = np.where(nudge < 0, np.nextafter(base, np.float32(-1e9)), np.where(nudge > 0, np.nextafter(base, np.float32(1e9)), base)).astype(np.float32)
        pts3[:, 2] += np.float32(0.5)
        pts = np.zeros((len(pts3), 4), np.float32); pts[:, :3] = pts3
        pts[:, 3] = rng.integers(0, 2**24, len(pts)).astype(np.uint32).view(np.float32)
        prm = P.launch_params(voxel_res=res, seed_res=res * 8)
        rc, olab, ores, oh = oracle.segment(pts, prm)
        assert rc == 0
        rc2, elab, eres, eh = emul.segment(pts, prm)
        assert rc2 == 0 and np.array_equal(olab, elab)
        glab = gpu_ctx.segment(pts, prm)
        assert gpu_ctx.result.octree_depth == ores.octree_depth
        for k in ("VOXEL_KEYS", "VOXEL_COUNT"):
            assert same_bits(oh.get(k), gpu_ctx.debug(k)), (res, k)
        assert np.array_equal(olab, glab), res
    assert ores.octree_depth > 10


def test_device_pointers_and_streams(P, oracle, gpu_ctx):
    """Caller-owned device buffers (torch tensors) in, labels out on the device, on a torch stream."""
    torch = pytest.importorskip("torch")
    pts = case_points(P, "rgbd_160x120"); prm = case_params(P, "rgbd_160x120")
    d_pts = torch.from_numpy(pts).cuda(); d_lab = torch.empty(len(pts), dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    gpu_ctx.set_stream(st.cuda_stream)
    gpu_ctx.segment(d_pts.data_ptr(), prm, labels_out=d_lab.data_ptr(), n=len(pts), on_device=True)
    gpu_ctx.set_stream(0)
    rc, olab, _, _ = oracle.segment(pts, prm)
    assert np.array_equal(d_lab.cpu().numpy().view(np.uint32), olab)
    # host frames with the caller's own label array (reused from call to call)
    mine = np.full(len(pts), 0xDEADBEEF, np.uint32)
    assert gpu_ctx.segment(pts, prm, labels_out=mine) is mine and np.array_equal(mine, olab)
    with pytest.raises(ValueError):
        gpu_ctx.segment(pts, prm, labels_out=np.empty(len(pts) + 1, np.uint32))
    with pytest.raises(ValueError):
        gpu_ctx.segment(pts, prm, labels_out=np.empty(len(pts), np.int64))


def test_cli_on_fixture(P, oracle, tmp_path):
    import subprocess
    from conftest import FIXTURE_PCD
    exe = os.path.join(ROOT, "fast-3d-pointcloud-segmentation_amd", "supervoxel_clustering")
    out = str(tmp_path / "seg.pcd"); lab = str(tmp_path / "labels.u32")
    r = subprocess.run([exe, "--CVX", "--AL", "-t", "0.2", "-p", FIXTURE_PCD, "-o", out, "--labels", lab], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    assert "Found 597 supervoxels" in r.stdout
    pts = P.read_pcd(FIXTURE_PCD)
    rc, olab, ores, oh = oracle.segment(pts, P.launch_params())
    assert np.array_equal(np.fromfile(lab, np.uint32), olab)
    cloud, clab = P.read_pcd(out, with_labels=True)
    ox, ol, oc = oh.voxel_cloud()
    assert np.array_equal(cloud[:, :3].view(np.uint32), ox.view(np.uint32)) and np.array_equal(clab, ol)
    assert np.array_equal(cloud[:, 3].copy().view(np.uint32), oc)


def test_cli_refine_flag(P, oracle, tmp_path):
    """--refine 3 (main() calls refineSupervoxels(3, ...), :371): the count it prints and the labelled voxel cloud it writes
    are the oracle's."""
    import subprocess
    from conftest import FIXTURE_PCD
    exe = os.path.join(ROOT, "fast-3d-pointcloud-segmentation_amd", "supervoxel_clustering")
    out = str(tmp_path / "seg.pcd")
    r = subprocess.run([exe, "--CVX", "--AL", "-t", "0.2", "-p", FIXTURE_PCD, "-o", out, "--refine", "3"], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    pts = P.read_pcd(FIXTURE_PCD)
    rc, olab, ores, oh = oracle.segment(pts, P.launch_params())
    want = oh.refine(3)
    assert "Refining supervoxels...\n%d supervoxels after 3 refinement iterations" % len(want["label"]) in r.stdout
    cloud, lab = P.read_pcd(out + ".refined", with_labels=True)
    assert np.array_equal(lab, want["voxel_label"])
    xyz = oh.get("VOXEL_XYZ").reshape(-1, 3)
    assert np.array_equal(cloud[:, :3].view(np.uint32), xyz.view(np.uint32))


def test_batch_of_frames_matches_single_calls(P, oracle):
    """f3ds_segment_batch: one context per frame, one merge dispatch for all of them."""
    names = ["rgbd_160x120", "rgbd_320x240_ghosts", "fixture_launch_flags", "rgbd_160x120"]
    prm = P.launch_params(voxel_res=0.012, seed_res=0.1)
    frames = [case_points(P, n) for n in names] + [np.zeros((0, 4), np.float32)]
    ctxs = [P.Context(0) for _ in frames]
    got = P.segment_batch(ctxs, frames, prm)
    for f, g, c in zip(frames, got, ctxs):
        rc, olab, ores, _ = oracle.segment(f, prm)
        assert rc == 0 and np.array_equal(olab, g)
        assert c.result.n_regions == ores.n_regions and c.result.n_merges == ores.n_merges
    for c in ctxs:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["rgbd_160x120", "rgbd_320x240_ghosts", "rgbd_160x120_equalization", "fused_200k_nan_lambda", "fixture_launch_flags"])
def test_evaluation_scores_match_oracle(P, oracle, gpu_ctx, name):
    """f3ds_evaluate == the oracle's Testing::eval_performance, all seven floats bit for bit."""
    from golden_cases import synthetic_truth
    pts = case_points(P, name); prm = case_params(P, name)
    truth = np.zeros(len(pts), np.uint32) if name == "fixture_launch_flags" else synthetic_truth(pts)
    gpu_ctx.segment(pts, prm)
    rc, _, _, h = oracle.segment(pts, prm)
    rc, want = h.evaluate(truth)
    assert rc == 0
    assert gpu_ctx.evaluate(truth).as_dict() == want.as_dict()


@pytest.mark.gpu
def test_auto_threshold_matches_oracle(P, oracle, gpu_ctx):
    from golden_cases import synthetic_truth
    for name, sweep in (("rgbd_160x120", (0.05, 0.6, 0.05)), ("rgbd_320x240_ghosts", (0.0, 1.0, 0.125)), ("rgbd_160x120_manual_lambda", (0.8, 1.0, 0.005))):
        pts = case_points(P, name); prm = case_params(P, name)
        truth = synthetic_truth(pts)
        gpu_ctx.segment(pts, prm)
        bt, bp, table, labels = gpu_ctx.auto_threshold(prm, truth, *sweep)
        rc, _, _, h = oracle.segment(pts, prm)
        rc, obt, obp, otable, olabels = h.auto_threshold(prm, truth, len(pts), *sweep)
        assert rc == 0 and bt == obt and bp.as_dict() == obp.as_dict() and table == otable and np.array_equal(labels, olabels)
        assert gpu_ctx.result.n_regions == len(np.unique(olabels[olabels != 0xFFFFFFFF]))
        assert gpu_ctx.evaluate(truth).as_dict() == bp.as_dict() or bt == 0.0
    with pytest.raises(IndexError):                 # std::out_of_range, clustering.cpp:694-698
        gpu_ctx.auto_threshold(prm, truth, -0.5, 0.5, 0.1)
    with pytest.raises(IndexError):
        gpu_ctx.auto_threshold(prm, truth, 0.1, 0.5, 0.0)
    with pytest.raises(P.LogicError):
        P.Context(0).evaluate(np.zeros(0, np.uint32))


@pytest.mark.gpu
def test_clustering_mirror_all_thresh(P, oracle, gpu_ctx):
    from golden_cases import synthetic_truth
    pts = case_points(P, "rgbd_160x120"); truth = synthetic_truth(pts)
    sv = P.SupervoxelClustering(0.02, 0.2, context=gpu_ctx); sv.setInputCloud(pts)
    seg = P.Clustering(); seg.set_delta_g(P.CONVEX_NORMALS_DIFF); seg.set_initialstate(sv)
    table = seg.all_thresh(truth, 0.1, 0.5, 0.1)
    bt, bp = seg.best_thresh(table)
    seg.cluster(bt)
    rc, _, _, h = oracle.segment(pts, case_params(P, "rgbd_160x120"))
    rc, obt, obp, otable, olabels = h.auto_threshold(case_params(P, "rgbd_160x120"), truth, len(pts), 0.1, 0.5, 0.1)
    assert table == otable and bt == obt and np.array_equal(seg.get_point_labels(), olabels)
    assert seg.eval_performance(truth) == obp.as_dict()


@pytest.mark.gpu
def test_cli_automatic_threshold(P, oracle, tmp_path):
    """Without -t the tool runs all_thresh(0.8, 1, 0.005) + best_thresh and writes the seven <name>_*.csv files."""
    import subprocess
    from golden_cases import synthetic_truth
    pts = case_points(P, "rgbd_160x120"); truth = synthetic_truth(pts)
    src = str(tmp_path / "frame.pcd"); lab = str(tmp_path / "labels.u32")
    P.write_pcd(src, pts[:, :3], pts[:, 3].copy().view(np.uint32), truth)
    exe = os.path.join(ROOT, "fast-3d-pointcloud-segmentation_amd", "supervoxel_clustering")
    r = subprocess.run([exe, "--CVX", "--AL", "-v", "0.02", "-s", "0.2", "-p", src, "--labels", lab, "-f", "sweep"], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    prm = P.launch_params(voxel_res=0.02, seed_res=0.2)
    pts2, truth2 = P.read_pcd(src, with_labels=True)
    rc, _, _, h = oracle.segment(pts2, prm)
    rc, obt, obp, otable, olabels = h.auto_threshold(prm, truth2, len(pts2), 0.8, 1.0, 0.005)
    assert rc == 0 and "Using best threshold: %f (F-score %f, voi %f)" % (obt, obp.fscore, obp.voi) in r.stdout
    assert np.array_equal(np.fromfile(lab, np.uint32), olabels)
    row = open(str(tmp_path / "sweep_fscore.csv")).read().strip().rstrip(";").split(";")
    assert len(row) == len(otable) and [float(x) for x in row] == pytest.approx([s["fscore"] for s in otable.values()], rel=1e-5)
    assert "F-score\t%f" % obp.fscore in r.stdout or obt == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("shift", ["-1", "32", "2"])
def test_dirty_tile_sweeps_are_bit_identical_at_any_setting(P, shift):
    """F3DS_INC_SHIFT=-1 never skips a tile, 32 always tries to (falling back to the chain walker when the R rounds do
    not converge), 2 switches early: the sweep results must not depend on it.  The library reads the variable when
    it is loaded, hence the child process."""
    import hashlib, json, subprocess, sys
    names = ["rgbd_320x240_ghosts", "rgbd_320x240_large_supervoxels", "fixture_launch_flags"]
    code = (
        "import sys, json, hashlib; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import conftest; from golden_cases import case_points, case_params\n"
        "P = conftest.pkg(); ctx = P.Context(0); out = {}\n"
        "for n in %r:\n"
        "    lab = ctx.segment(case_points(P, n), case_params(P, n))\n"
        "    out[n] = dict(labels=conftest.sha_of(lab), **{w: conftest.sha_of(ctx.debug(w)) for w in ('VOXEL_SVLABEL', 'VOXEL_DIST', 'SV_CENTROID', 'MERGES')})\n"
        "print(json.dumps(out))\n") % (ROOT, os.path.join(ROOT, "tests"), names)
    env = dict(os.environ, F3DS_INC_SHIFT=shift)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_golden.json")))
    for n in names:
        assert got[n]["labels"] == gold[n]["labels_sha256"], (n, shift)
        for w in ("VOXEL_SVLABEL", "VOXEL_DIST", "SV_CENTROID", "MERGES"):
            assert got[n][w] == gold[n]["sha256"][w], (n, w, shift)


@pytest.mark.gpu
@pytest.mark.parametrize("rounds", ["1", "2"])
def test_incremental_R_rounds_that_do_not_converge_fall_back_to_the_chain_walker(P, rounds):
    """An incremental sweep whose last R round still changes ownR turns full (d_sweep_R_round); no pre-pass has run for it, so pass 0 of the chain
    walker derives R for every voxel itself (d_sweep_R, `sweep_pre`).  Until round 5 that fallback did nothing -- the walker found an empty work list
    and the claim pass ran on a half-converged ownR -- and no test reached it: three rounds settle every frame seen so far.  F3DS_R_ROUNDS_RUN=1 makes
    round 0 the last one (any change at all falls back), F3DS_INC_SHIFT=32 makes every sweep from the third on incremental."""
    import json, subprocess, sys
    names = ["rgbd_320x240_ghosts", "rgbd_320x240_large_supervoxels", "fixture_launch_flags", "rgbd_160x120"]
    code = (
        "import sys, json, hashlib; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import conftest; from golden_cases import case_points, case_params\n"
        "P = conftest.pkg(); ctx = P.Context(0); out = {}\n"
        "for n in %r:\n"
        "    lab = ctx.segment(case_points(P, n), case_params(P, n))\n"
        "    out[n] = dict(labels=conftest.sha_of(lab), **{w: conftest.sha_of(ctx.debug(w)) for w in ('VOXEL_SVLABEL', 'VOXEL_DIST', 'SV_CENTROID', 'MERGES')})\n"
        "    out[n]['stats'] = ctx.sweep_stats(); out[n]['sweeps'] = int(ctx.result.sweeps)\n"
        "    r = ctx.refine_supervoxels(2); out[n]['refined'] = conftest.sha_of(r['voxel_label'])\n"
        "print(json.dumps(out))\n") % (ROOT, os.path.join(ROOT, "tests"), names)
    got = {}
    for env_rounds in (rounds, None):
        env = dict(os.environ, F3DS_INC_SHIFT="32")
        if env_rounds:
            env["F3DS_R_ROUNDS_RUN"] = env_rounds
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        got[env_rounds] = json.loads(r.stdout.strip().splitlines()[-1])
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_golden.json")))
    for n in names:
        assert got[rounds][n]["labels"] == gold[n]["labels_sha256"], (n, rounds)
        for w in ("VOXEL_SVLABEL", "VOXEL_DIST", "SV_CENTROID", "MERGES"):
            assert got[rounds][n][w] == gold[n]["sha256"][w], (n, w, rounds)
        assert got[rounds][n]["refined"] == got[None][n]["refined"], (n, "refineSupervoxels")
        for k in (rounds, None):      # F3DS_DBG_SWEEP_STATS: every sweep is one of full / incremental / fallback
            assert sum(got[k][n]["stats"]) == got[k][n]["sweeps"] and got[k][n]["stats"][1] + got[k][n]["stats"][2] > 0, (n, k, got[k][n]["stats"])
    # the point of the test: the fallback is TAKEN (final hashes alone do not prove that) -- with one round on every case (two rounds settle most small frames)
    fallbacks = [got[rounds][n]["stats"][2] for n in names]
    if rounds == "1":
        assert all(f > 0 for f in fallbacks), fallbacks


MERGE_VARIANTS = ([dict(F3DS_MERGE_NW=nw, F3DS_MERGE_KEYS=k) for nw in ("4", "8") for k in ("lds", "global")] + [dict(F3DS_FORCE_GLOBAL_MERGE="1")] +
                  [dict(F3DS_MERGE_NW=nw, F3DS_MERGE_KEYS=k, F3DS_MERGE_SPEC="0") for nw in ("4", "8") for k in ("lds", "global")])      # (the loops commit two merges per epoch where they can: the last variants switch that off)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", MERGE_VARIANTS, ids=lambda v: "-".join("%s" % x for x in v.values()))
def test_every_merge_kernel_layout_gives_the_oracle_merges(P, oracle, monkeypatch, variant):
    """The merge loop exists as d_merge_il_t<4 | 8 waves, per-edge arrays in LDS | global memory> (chosen by what else runs on the device and by what fits
    LDS; each with and without the speculative second merge of an epoch) and as the all-global d_merge: each forced here (switches are read per call)
    on golden cases and on the 1M-point frame (regions of > 30 000 voxels and hundreds of leaves: multi-chunk staging)."""
    for k, v in variant.items():
        monkeypatch.setenv(k, v)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_golden.json")))
    big = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_golden_big.json")))
    ctx = P.Context(0)
    for n in ["rgbd_320x240_ghosts", "rgbd_160x120_equalization", "fixture_launch_flags", "fused_200k_nan_lambda", "rgbd_320x240_large_supervoxels",
              "rgbd_160x120_threshold_1", "rgbd_160x120_rgb_metric"]:
        lab = ctx.segment(case_points(P, n), case_params(P, n))
        assert sha_of(lab) == gold[n]["labels_sha256"], n
        assert sha_of(ctx.debug("MERGES")) == gold[n]["sha256"]["MERGES"], n
    # voxel cloud order (leaf arrays of the merged regions) and region records against the oracle run here
    pts = case_points(P, "rgbd_320x240_ghosts"); prm = case_params(P, "rgbd_320x240_ghosts")
    rc, olab, ores, oh = oracle.segment(pts, prm)
    assert np.array_equal(ctx.segment(pts, prm), olab)
    ox, ol, _ = oh.voxel_cloud(); gx, gl, _ = ctx.voxel_cloud()
    assert np.array_equal(ox.view(np.uint32), gx.view(np.uint32)) and np.array_equal(ol, gl)
    for seed in (1000, 1061):
        e = big["config5_seed%d" % seed]
        lab = ctx.segment(P.synth_frame(*e["synth"]), P.launch_params(**e["params"]))
        assert sha_of(lab) == e["labels_sha256"], seed
        assert sha_of(ctx.debug("MERGES")) == e["sha256"]["MERGES"], seed
    # supervoxels of ~630 voxels (seed / voxel resolution 60): the speculative second merge of the 8-wave layout absorbs regions of up to 1024 rows,
    # staged in two rounds of one row per lane
    pts = P.synth_frame(0, 77, 640, 480, 10); prm = P.launch_params(voxel_res=0.005, seed_res=0.3)
    rc, olab, ores, oh = oracle.segment(pts, prm)
    assert rc == 0 and ores.n_merges > 100 and ores.n_voxels > 512 * ores.n_supervoxels
    assert np.array_equal(ctx.segment(pts, prm), olab)
    assert np.array_equal(ctx.debug("MERGES"), oh.get("MERGES"))
    ctx.close()


@pytest.mark.gpu
def test_vccs_getters_match_oracle(P, oracle, gpu_ctx):
    """getVoxelCentroidCloud / supervoxel_clusters / makeSupervoxelNormalCloud / getSupervoxelAdjacency
    (src/supervoxel_clustering.cpp:356-365) through the public accessors and the SupervoxelClustering mirror."""
    pts = case_points(P, "rgbd_320x240_ghosts"); prm = case_params(P, "rgbd_320x240_ghosts")
    sv = P.SupervoxelClustering(prm.voxel_res, prm.seed_res, context=gpu_ctx); sv.setInputCloud(pts)
    clusters = sv.extract()
    rc, _, ores, h = oracle.segment(pts, prm)
    assert rc == 0
    xyz, rgba = sv.getVoxelCentroidCloud(); lxyz, lab = sv.getLabeledVoxelCloud()
    assert np.array_equal(xyz.view(np.uint32), h.get("VOXEL_XYZ").reshape(-1, 3).view(np.uint32)) and np.array_equal(lab, h.get("VOXEL_SVLABEL"))
    rgb = h.get("VOXEL_RGB").reshape(-1, 3).astype(np.uint32)
    assert np.array_equal(rgba, (rgb[:, 0] << 16) | (rgb[:, 1] << 8) | rgb[:, 2])
    cen = h.get("SV_CENTROID").reshape(-1, 10)
    assert np.array_equal(clusters["label"], h.get("SV_LABELS")) and len(clusters["label"]) == ores.n_supervoxels
    assert np.array_equal(clusters["xyz"].view(np.uint32), cen[:, 0:3].copy().view(np.uint32))
    assert np.array_equal(clusters["rgb"].view(np.uint32), cen[:, 3:6].copy().view(np.uint32))
    nxyz, nrm = sv.makeSupervoxelNormalCloud()
    assert np.array_equal(nrm.view(np.uint32), cen[:, 6:9].copy().view(np.uint32))
    assert clusters["n_voxels"].sum() >= int((lab != 0).sum())             # every owned voxel is a leaf (a ghost leaf counts twice)
    assert np.array_equal(sv.getSupervoxelAdjacency(), h.get("EDGES").reshape(-1, 2))


@pytest.mark.gpu
def test_random_small_frames_and_parameters(P, oracle, gpu_ctx):
    """Seeded sweep over frame sizes, NaN rates, resolutions, metrics, merging modes, thresholds, leaf orders and
    importances: every intermediate array of the device path equals the oracle's, byte for byte."""
    rng = np.random.default_rng(int(os.environ.get("F3DS_FUZZ_SEED", "20260930")))
    ran = 0
    for it in range(int(os.environ.get("F3DS_FUZZ_CASES", "24"))):
        w, hgt = int(rng.integers(20, 260)), int(rng.integers(20, 200))
        kind = int(rng.integers(0, 2))
        pts = P.synth_frame(kind, int(rng.integers(1, 10**6)), w, hgt, int(rng.integers(0, 400)) if kind == 0 else 0)
        vres = float(rng.choice([0.008, 0.01, 0.015, 0.02, 0.03, 0.05, 0.08]))
        prm = P.launch_params(voxel_res=vres, seed_res=vres * float(rng.choice([2, 3, 5, 8, 12, 20])),
                              use_transform=int(rng.integers(0, 2)) if kind == 0 else 0, color_metric=int(rng.integers(0, 2)),
                              geom_metric=int(rng.integers(0, 2)), merging=int(rng.integers(0, 3)), lambda_=float(rng.uniform(0.0, 1.0)),
                              bins=int(rng.choice([0, 5, 20, 100, 500])), threshold=float(rng.choice([0.0, 0.05, 0.1, 0.3, 0.6, 1.0])),
                              leaf_order=int(rng.integers(0, 2)), w_color=float(rng.uniform(0.0, 1.0)), w_spatial=float(rng.uniform(0.0, 1.0)),
                              w_normal=float(rng.uniform(0.0, 6.0)))
        rc, olab, ores, oh = oracle.segment(pts, prm)
        if rc != 0:                      # the same refusal on both sides (e.g. the equalization bin overrun)
            with pytest.raises(Exception):
                gpu_ctx.segment(pts, prm)
            continue
        lab = gpu_ctx.segment(pts, prm)
        assert np.array_equal(lab, olab), (it, first_mismatch(lab, olab))
        for wname in ALL_DEBUG:
            a, b = gpu_ctx.debug(wname), oh.get(wname)
            assert same_bits(a, b), (it, wname)
        ran += 1
    assert ran >= 16


@pytest.mark.gpu
def test_random_recluster_evaluation_and_batches(P, oracle):
    """Seeded sweep over the other entry points: f3ds_recluster with changed metrics, f3ds_evaluate and
    f3ds_auto_threshold with random ground truth, and f3ds_segment_batch over ragged mixes of frames."""
    rng = np.random.default_rng(int(os.environ.get("F3DS_FUZZ_SEED", "777")))
    ncase = int(os.environ.get("F3DS_FUZZ_CASES", "8"))
    ctx = P.Context(0)
    for it in range(ncase):
        w, hgt = int(rng.integers(60, 220)), int(rng.integers(50, 160))
        pts = P.synth_frame(0, int(rng.integers(1, 10**6)), w, hgt, int(rng.integers(0, 200)))
        vres = float(rng.choice([0.012, 0.02, 0.03]))
        prm = P.launch_params(voxel_res=vres, seed_res=vres * float(rng.choice([4, 8, 12])), threshold=float(rng.choice([0.1, 0.3])))
        ctx.segment(pts, prm)
        rc, olab, ores, oh = oracle.segment(pts, prm)
        assert rc == 0
        # recluster with other metrics / threshold
        p2 = P.launch_params(voxel_res=prm.voxel_res, seed_res=prm.seed_res, color_metric=int(rng.integers(0, 2)), geom_metric=int(rng.integers(0, 2)),
                             merging=int(rng.integers(0, 2)), lambda_=float(rng.uniform(0.1, 0.9)), threshold=float(rng.choice([0.05, 0.2, 0.5, 1.0])))
        l2 = ctx.recluster(p2)
        rc, ol2, _ = oh.cluster(p2, len(pts))
        assert rc == 0 and np.array_equal(l2, ol2), it
        # evaluation against random ground truth (blocks of the image plus noise)
        truth = ((np.arange(len(pts)) // w // 16) * 7 + (np.arange(len(pts)) % w) // 16).astype(np.uint32) % int(rng.integers(3, 400))
        truth[rng.random(len(pts)) < 0.01] = int(rng.integers(0, 1000))
        rc, want = oh.evaluate(truth)
        assert rc == 0 and ctx.evaluate(truth).as_dict() == want.as_dict(), it
        sweep = (float(rng.choice([0.0, 0.1])), float(rng.choice([0.5, 1.0])), float(rng.choice([0.05, 0.125])))
        bt, bp, table, labels = ctx.auto_threshold(p2, truth, *sweep)
        rc, obt, obp, otable, olabels = oh.auto_threshold(p2, truth, len(pts), *sweep)
        assert rc == 0 and bt == obt and table == otable and np.array_equal(labels, olabels), it
    ctx.close()
    # batches: ragged sizes, an empty frame, an all-NaN frame, duplicates
    for it in range(max(2, ncase // 4)):
        frames = []
        for k in range(int(rng.integers(2, 7))):
            kind = int(rng.integers(0, 4))
            if kind == 0:
                frames.append(np.zeros((0, 4), np.float32))
            elif kind == 1:
                f = P.synth_frame(0, int(rng.integers(1, 10**6)), 40, 30, 0); f[:, :3] = np.nan; frames.append(f)
            else:
                frames.append(P.synth_frame(int(rng.integers(0, 2)), int(rng.integers(1, 10**6)), int(rng.integers(30, 200)), int(rng.integers(30, 150)), 50 * (kind == 2)))
        prm = P.launch_params(voxel_res=0.02, seed_res=float(rng.choice([0.1, 0.2])), use_transform=0)
        ctxs = [P.Context(0) for _ in frames]
        got = P.segment_batch(ctxs, frames, prm)
        for f, g, c in zip(frames, got, ctxs):
            rc, olab, ores, _ = oracle.segment(f, prm)
            assert rc == 0 and np.array_equal(olab, g), it
            assert c.result.n_regions == ores.n_regions and c.result.n_voxels == ores.n_voxels
        for c in ctxs:
            c.close()


@pytest.mark.gpu
def test_degenerate_clouds(P, oracle):
    """Empty / one-point / duplicate / collinear / planar / NaN / inf / negative-z clouds with random parameters
    (tools/fuzz_clouds.py): the device path and the oracle agree on the return code and on every array."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_clouds.py"), "80", "9", "--gpu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "mismatches 0" in r.stdout.splitlines()[-1], r.stdout[-2000:]


@pytest.mark.gpu
def test_cli_flags_and_directory_mode(P, oracle, tmp_path):
    """The reference's flags map onto the same parameters as the API: -d over two files, -r, --NT, --RGB, --ML / --EQ."""
    import subprocess
    from golden_cases import synthetic_truth
    exe = os.path.join(ROOT, "fast-3d-pointcloud-segmentation_amd", "supervoxel_clustering")
    d = tmp_path / "frames"; d.mkdir()
    frames = {}
    for name, seed in (("a", 5), ("b", 6)):
        pts = P.synth_frame(0, seed, 160, 120, 20)
        truth = synthetic_truth(pts) % 7
        P.write_pcd(str(d / (name + ".pcd")), pts[:, :3], pts[:, 3].copy().view(np.uint32), truth)
        frames[name] = P.read_pcd(str(d / (name + ".pcd")), with_labels=True)
    combos = [
        (["--NT", "--RGB", "--ML", "0.3", "-t", "0.25", "-v", "0.02", "-s", "0.2", "-c", "0.5", "-z", "0.3", "-n", "2.0"],
         dict(use_transform=0, color_metric=1, geom_metric=0, merging=0, lambda_=0.3, threshold=0.25, voxel_res=0.02, seed_res=0.2, w_color=0.5, w_spatial=0.3, w_normal=2.0), None),
        (["--CVX", "--EQ", "50", "-t", "0.5", "-v", "0.02", "-s", "0.2"],
         dict(geom_metric=1, merging=2, bins=50, threshold=0.5, voxel_res=0.02, seed_res=0.2), None),
        (["--CVX", "--AL", "-t", "0.2", "-v", "0.02", "-s", "0.2", "-r", "3"], dict(threshold=0.2, voxel_res=0.02, seed_res=0.2), 3),
    ]
    for flags, kw, removed in combos:
        lab = str(tmp_path / "lab")
        r = subprocess.run([exe, "-d", str(d)] + flags + ["--labels", lab, "-f", "res"], capture_output=True, text=True, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr
        assert "Found 2 files" in r.stdout and "Average scores" in r.stdout
        prm = P.default_params(**dict(dict(color_metric=0, geom_metric=0, merging=1), **kw)); prm.fold_negative_z = 1
        for name, (pts, truth) in frames.items():
            if removed is not None:                           # main():332-336: drop the label and the NaN points
                keep = (truth != removed) & ~np.isnan(pts[:, 2])
                pts = pts[keep]
            rc, olab, _, _ = oracle.segment(pts, prm)
            assert rc == 0 and np.array_equal(np.fromfile(lab + "." + name, np.uint32), olab), (flags, name)
    r = subprocess.run([exe, "-p", str(d / "a.pcd"), "--ML", "--AL"], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 1 and "Only one parameter" in r.stderr
    # --stream: the same files through the frame pipeline, label files only
    lab = str(tmp_path / "slab")
    r = subprocess.run([exe, "-d", str(d), "--CVX", "--AL", "-t", "0.2", "-v", "0.02", "-s", "0.2", "--labels", lab, "--stream", "2"], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    prm = P.launch_params(voxel_res=0.02, seed_res=0.2); prm.fold_negative_z = 1
    for name, (pts, truth) in frames.items():
        rc, olab, ores, _ = oracle.segment(pts, prm)
        assert np.array_equal(np.fromfile(lab + "." + name, np.uint32), olab), name
        assert "%s.pcd: %d points, %d voxels" % (name, len(pts), ores.n_voxels) in r.stdout
    r = subprocess.run([exe, "-d", str(d), "--stream", "2", "--labels", lab], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 1 and "--stream needs -t" in r.stderr


def test_frame_pipeline_returns_every_frame_in_order_with_the_labels_of_single_calls(P, oracle, gpu_ctx):
    """Row N4 (f3ds_stream_*): frames of different sizes and two parameter sets through a pipeline of depth 4 with two
    worker threads; tags come back in submission order, labels and counts are those of f3ds_segment on the same
    frame, which the oracle checks for a few of them."""
    frames = [P.synth_frame(0, 4000 + i, 96 + 16 * (i % 3), 72 + 8 * (i % 2), 25) for i in range(11)]
    frames[4] = np.zeros((0, 4), np.float32)                     # an empty frame in the middle
    frames[7] = np.full((50, 4), np.float32("nan"))              # and one without a finite point
    prms = [P.launch_params(voxel_res=0.02, seed_res=0.2), P.launch_params(voxel_res=0.03, seed_res=0.2, threshold=0.3)]
    pick = lambda i: prms[1 if i in (2, 3, 8) else 0]
    want = []
    for i, f in enumerate(frames):
        lab = gpu_ctx.segment(f, pick(i))
        want.append((lab, gpu_ctx.result.n_voxels, gpu_ctx.result.n_regions))
    for i in (0, 3, 10):
        rc, olab, ores, _ = oracle.segment(frames[i], pick(i))
        assert rc == 0 and np.array_equal(olab, want[i][0])
    with P.FrameStream(0, depth=4, groups=2) as fs:
        assert fs.next() is None and fs.pending() == 0           # nothing in flight
        got = []
        i = 0
        while i < len(frames):
            if fs.submit(frames[i], pick(i), tag=100 + i):
                i += 1
            else:
                assert fs.pending() == 4                         # full: never blocks, take the oldest
                got.append(fs.next())
        while fs.pending():
            got.append(fs.next())
        assert fs.next() is None
    assert [g[0] for g in got] == [100 + i for i in range(len(frames))]
    for i, (tag, lab, res) in enumerate(got):
        assert np.array_equal(lab, want[i][0]), i
        assert (res.n_points, res.n_voxels, res.n_regions) == (len(frames[i]), want[i][1], want[i][2]), i


def test_frame_pipeline_run_generator(P, gpu_ctx):
    """FrameStream.run: an iterable of frames in, (index, labels, Result) out in order, with more frames than slots."""
    prm = P.launch_params(voxel_res=0.02, seed_res=0.2)
    frames = [P.synth_frame(0, 6000 + i, 80 + 8 * (i % 4), 60, 10) for i in range(9)]
    want = [gpu_ctx.segment(f, prm) for f in frames]
    with P.FrameStream(0, depth=3) as fs:
        got = list(fs.run(iter(frames), prm))
    assert [g[0] for g in got] == list(range(9))
    for i, (_, lab, res) in enumerate(got):
        assert np.array_equal(lab, want[i]) and res.n_points == len(frames[i]), i


def test_frame_pipeline_can_be_destroyed_with_frames_in_flight(P, gpu_ctx):
    """f3ds_stream_destroy lets the workers finish the batch they are in and drops what was never taken; a new pipeline
    (and the plain context) work afterwards."""
    prm = P.launch_params(voxel_res=0.02, seed_res=0.2)
    f0 = P.synth_frame(0, 7000, 160, 120, 10)
    w0 = gpu_ctx.segment(f0, prm)
    fs = P.FrameStream(0, depth=6, groups=2)
    for i in range(6):
        assert fs.submit(f0, prm, i)
    fs.close()                                                   # nothing taken
    fs.close()                                                   # idempotent
    with P.FrameStream(0, depth=2) as fs2:
        assert fs2.submit(f0, prm, 1)
        assert np.array_equal(fs2.next()[1], w0)
    assert np.array_equal(gpu_ctx.segment(f0, prm), w0)


def test_frame_pipeline_buffers_capacity_and_threads(P, gpu_ctx):
    """Zero-copy submission through the slot's pinned buffer, F3DS_ERR_CAPACITY leaves the frame in place, a frame that
    fails (voxel grid too deep) reports its own status without disturbing its neighbours, and a producer thread
    feeding while this thread consumes ends with every frame delivered once."""
    import ctypes, threading, time
    prm = P.launch_params(voxel_res=0.02, seed_res=0.2)
    f0 = P.synth_frame(0, 5000, 128, 96, 20)
    w0 = gpu_ctx.segment(f0, prm)
    lib = P.load_library()
    with P.FrameStream(0, depth=3, groups=3) as fs:
        buf = fs.buffer(len(f0))
        buf[:] = f0
        assert fs.submit(buf, prm, 7)
        n = ctypes.c_size_t(); tag = ctypes.c_uint64()
        small = np.empty(10, np.uint32)
        assert lib.f3ds_stream_next(fs.handle, small.ctypes.data, 10, ctypes.byref(n), ctypes.byref(tag), None, 1) == P.ERR_CAPACITY
        assert n.value == len(f0) and tag.value == 7 and fs.pending() == 1
        t, lab, res = fs.next()
        assert t == 7 and np.array_equal(lab, w0)
        assert fs.peek() is None and fs.submit(f0, prm, 8)
        t, view, res = fs.peek()                                 # labels in place, frame still in the pipeline
        assert t == 8 and np.array_equal(view, w0) and fs.pending() == 1 and res.n_points == len(f0)
        fs.drop()
        assert fs.pending() == 0
        # a frame the device path refuses (grid deeper than 21 levels) between two good ones, same parameters: whether
        # the worker ran them as one batch or not, each frame reports its own status
        bad = f0.copy(); bad[0, :3] = (1e6, 1e6, 5.0)
        assert fs.submit(f0, prm, 1) and fs.submit(bad, prm, 2) and fs.submit(f0, prm, 3)
        assert not fs.submit(f0, prm, 4) and fs.buffer(10) is None
        assert np.array_equal(fs.next()[1], w0)
        with pytest.raises(P.F3dsError) as e:
            fs.next()
        assert e.value.code == P.ERR_DEPTH
        assert np.array_equal(fs.next()[1], w0) and fs.pending() == 0
    with P.FrameStream(0, depth=3, groups=1) as fs:              # one worker: the three frames can share a batch call
        assert fs.submit(f0, prm, 1) and fs.submit(bad, prm, 2) and fs.submit(f0, prm, 3)
        assert np.array_equal(fs.next()[1], w0)
        with pytest.raises(P.F3dsError) as e:
            fs.next()
        assert e.value.code == P.ERR_DEPTH
        assert np.array_equal(fs.next()[1], w0) and fs.pending() == 0
    with P.FrameStream(0, depth=6, groups=2) as fs:
        # producer / consumer
        frames = [P.synth_frame(0, 5100 + i, 96, 72, 20) for i in range(4)]
        wants = [gpu_ctx.segment(f, prm) for f in frames]
        total = 24
        def produce():
            i = 0
            while i < total:
                if fs.submit(frames[i % 4], prm, i):
                    i += 1
                else:
                    time.sleep(0.0005)
        th = threading.Thread(target=produce); th.start()
        seen = 0
        while seen < total:
            r = fs.next(wait=True)
            if r is None:
                continue
            assert r[0] == seen and np.array_equal(r[1], wants[seen % 4]), seen
            seen += 1
        th.join()
        assert fs.pending() == 0


@pytest.mark.parametrize("name", ["rgbd_160x120", "rgbd_320x240_ghosts", "fixture_launch_flags", "rgbd_320x240_large_supervoxels"])
def test_refine_supervoxels_matches_oracle(P, oracle, gpu_ctx, name):
    """Row N3: refineSupervoxels(k) -- owned-two-ring normals, reseeding at the voxel nearest to each centroid, sweeps
    again -- bit for bit against the oracle's restatement, k = 0, 1 and the reference's 3 (src/supervoxel_clustering.cpp:371);
    the frame's own supervoxels, labels and a later recluster do not notice."""
    pts = case_points(P, name); prm = case_params(P, name)
    rc, olab, ores, oh = oracle.segment(pts, prm)
    glab = gpu_ctx.segment(pts, prm)
    assert rc == 0 and np.array_equal(olab, glab)
    with pytest.raises(P.F3dsError):
        P._check(gpu_ctx.lib, gpu_ctx.lib.f3ds_get_refined_voxels(gpu_ctx.handle, None, None, 0, None))      # before refine: logic error
    for k in (0, 1, 3):
        want = oh.refine(k); got = gpu_ctx.refine_supervoxels(k)
        for key in ("voxel_label", "label", "n_voxels", "voxel_normal", "xyz", "rgb", "normal"):
            assert want[key].shape == got[key].shape, (k, key)
            assert same_bits(want[key], got[key]), (k, key, first_mismatch(key, want[key], got[key]))
    assert not first_mismatch("VOXEL_SVLABEL", oh.get("VOXEL_SVLABEL"), gpu_ctx.debug("VOXEL_SVLABEL"))
    assert not first_mismatch("SV_CENTROID", oh.get("SV_CENTROID"), gpu_ctx.debug("SV_CENTROID"))
    assert not first_mismatch("VOXEL_NORMAL", oh.get("VOXEL_NORMAL"), gpu_ctx.debug("VOXEL_NORMAL"))
    assert np.array_equal(gpu_ctx.recluster(prm), olab)


def test_refine_supervoxels_random_frames(P, oracle, gpu_ctx):
    rng = np.random.default_rng(77)
    for i in range(6):
        w, h = int(rng.integers(40, 140)), int(rng.integers(30, 100))
        pts = P.synth_frame(0, 9000 + i, w, h, int(rng.integers(0, 80)))
        prm = P.launch_params(voxel_res=float(rng.choice([0.02, 0.03, 0.05])), seed_res=float(rng.choice([0.1, 0.2, 0.3])), use_transform=int(rng.integers(0, 2)))
        rc, olab, ores, oh = oracle.segment(pts, prm)
        glab = gpu_ctx.segment(pts, prm)
        assert rc == 0 and np.array_equal(olab, glab)
        k = int(rng.integers(1, 4))
        want = oh.refine(k); got = gpu_ctx.refine_supervoxels(k)
        for key in want:
            assert want[key].shape == got[key].shape and same_bits(want[key], got[key]), (i, k, key)


@pytest.mark.gpu
@pytest.mark.parametrize("force_rccl", [False, True])
def test_multi_gpu_driver_on_one_gpu(P, oracle, monkeypatch, force_rccl):
    """f3ds_multi_* (the C++ one-process multi-GPU batch driver) with one device: frames of mixed size, an empty one, results
    of single calls; a second call reuses contexts and blocks.  With F3DS_MULTI_FORCE_RCCL the label block also travels
    through a grouped ncclSend / ncclRecv pair (librccl loaded at run time), which is all of the exchange one GPU can show."""
    if force_rccl:
        monkeypatch.setenv("F3DS_MULTI_FORCE_RCCL", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    names = ["rgbd_160x120", "rgbd_320x240_ghosts", "fixture_launch_flags", "rgbd_160x120", "fused_200k_nan_lambda"]
    prm = P.launch_params(voxel_res=0.012, seed_res=0.1)
    frames = [case_points(P, n) for n in names] + [np.zeros((0, 4), np.float32)]
    mg = P.MultiGpu(n_devices=1, max_frames_per_device=8)
    assert mg.devices() == 1 and mg.device_of_frame(5) == 0
    for rep in range(2):
        use = frames if rep == 0 else frames[1:4]
        labels, results = mg.segment(use, prm)
        for f, g, r in zip(use, labels, results):
            rc, olab, ores, _ = oracle.segment(f, prm)
            assert rc == 0 and np.array_equal(olab, g)
            assert r.n_regions == ores.n_regions and r.n_merges == ores.n_merges and r.n_points == len(f)
    with pytest.raises(P.F3dsError) as e:
        mg.segment(frames * 2, prm)          # 12 frames > 1 device x 8
    assert e.value.code == P.ERR_CAPACITY
    mg.close()
    with pytest.raises(P.F3dsError):
        P.MultiGpu(n_devices=P.device_count() + 1)


@pytest.mark.gpu
def test_cli_gpus_and_dump(P, oracle, tmp_path):
    """--gpus 1 (directory of files through f3ds_multi_*) and --dump <dir> (what visualize() draws,
    /root/reference/src/supervoxel_clustering.cpp:586-700, as files) against the oracle's getters."""
    import subprocess
    exe = os.path.join(ROOT, "fast-3d-pointcloud-segmentation_amd", "supervoxel_clustering")
    d = tmp_path / "in"; d.mkdir()
    frames = {"a": P.synth_frame(0, 21, 160, 120, 30), "b": P.synth_frame(0, 22, 200, 150, 10), "c": P.synth_frame(1, 23, 100, 80, 0)}
    for k, f in frames.items():
        P.write_pcd(str(d / (k + ".pcd")), f[:, :3], f[:, 3].copy().view(np.uint32))
    flags = ["-v", "0.02", "-s", "0.2", "--CVX", "--AL", "-t", "0.2"]
    prm = P.launch_params(voxel_res=0.02, seed_res=0.2)
    r = subprocess.run([exe, "-d", str(d), "--gpus", "1", "--labels", str(tmp_path / "lab")] + flags, capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr + r.stdout
    for k, f in frames.items():
        rc, olab, ores, _ = oracle.segment(P.read_pcd(str(d / (k + ".pcd"))), prm)
        assert np.array_equal(np.fromfile(str(tmp_path / ("lab." + k)), np.uint32), olab), k
    # --dump on one file
    dump = tmp_path / "dump"
    r = subprocess.run([exe, "-p", str(d / "b.pcd"), "--dump", str(dump)] + flags, capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr + r.stdout
    pts = P.read_pcd(str(d / "b.pcd"))
    rc, olab, ores, oh = oracle.segment(pts, prm)
    cloud, sv = P.read_pcd(str(dump / "voxel_centroids.pcd"), with_labels=True)
    assert np.array_equal(cloud[:, :3].view(np.uint32), oh.get("VOXEL_XYZ").reshape(-1, 3).view(np.uint32)) and np.array_equal(sv, oh.get("VOXEL_SVLABEL"))
    cloud, lab = P.read_pcd(str(dump / "colored_voxels.pcd"), with_labels=True)
    ox, ol, oc = oh.voxel_cloud()
    assert np.array_equal(cloud[:, :3].view(np.uint32), ox.view(np.uint32)) and np.array_equal(lab, ol) and np.array_equal(cloud[:, 3].copy().view(np.uint32), oc)
    want = oh.refine(3)
    raw = open(str(dump / "supervoxel_normals.pcd"), "rb").read()
    body = np.frombuffer(raw[raw.index(b"DATA binary\n") + 12:], np.float32).reshape(-1, 6)
    assert np.array_equal(body[:, :3].view(np.uint32), want["xyz"].view(np.uint32)) and np.array_equal(body[:, 3:].view(np.uint32), want["normal"].view(np.uint32))
    # region adjacency: every initial adjacency mapped to the labels its supervoxels ended in (merge log replayed here)
    edges = oh.get("EDGES").reshape(-1, 2); merges = oh.get("MERGES").reshape(-1, 3)
    parent = {}
    def root(x):
        while x in parent:
            x = parent[x]
        return x
    for a, b, _ in merges:
        parent[int(b)] = int(a)
    exp = sorted({(min(root(int(a)), root(int(b))), max(root(int(a)), root(int(b)))) for a, b in edges if root(int(a)) != root(int(b))})
    rows = [l.split(",") for l in open(str(dump / "adjacency.csv")).read().splitlines()[1:]]
    assert [(int(x[0]), int(x[1])) for x in rows] == exp and len(exp) > 0
    cen = {int(l): c for l, c in zip(oh.get("SV_LABELS"), oh.get("SV_CENTROID").reshape(-1, 10)[:, :3])}
    for x in rows[:50]:
        assert np.allclose([float(v) for v in x[2:5]], cen[int(x[0])], rtol=0, atol=1e-6) and np.allclose([float(v) for v in x[5:8]], cen[int(x[1])], rtol=0, atol=1e-6)


@pytest.mark.gpu
def test_adjacency_list_regrows_instead_of_failing(P, monkeypatch):
    """More adjacencies than the list holds (S0 * 32 + 1024 by default) used to be F3DS_ERR_UNSUPPORTED; now the pass runs again with
    four times the room.  Forced here by starting with S0 * 1 + 1024 (F3DS_EDGE_MULT, read when a context is created)."""
    monkeypatch.setenv("F3DS_EDGE_MULT", "1")
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_golden.json")))
    big = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_golden_big.json")))
    ctx = P.Context(0)
    for n in ["rgbd_320x240_ghosts", "fixture_launch_flags"]:
        lab = ctx.segment(case_points(P, n), case_params(P, n))
        assert sha_of(lab) == gold[n]["labels_sha256"] and sha_of(ctx.debug("EDGES")) == gold[n]["sha256"]["EDGES"], n
    e = big["config5_seed1003"]
    frames = [P.synth_frame(*e["synth"]), case_points(P, "rgbd_160x120")]
    ctxs = [P.Context(0), P.Context(0)]
    labs = P.segment_batch(ctxs, frames, P.launch_params(**e["params"]))       # a batch in which only one frame overflows
    assert ctxs[0].result.n_edges > ctxs[0].result.n_seeds + 1024 and sha_of(labs[0]) == e["labels_sha256"]
    for c in ctxs + [ctx]:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("force_rccl", [False, True])
def test_multi_gpu_driver_pipelined_submit_collect(P, oracle, monkeypatch, force_rccl):
    """f3ds_multi_submit / f3ds_multi_collect: two batches in flight (batch k+1 computes while batch k is gathered and copied
    out), a third refused with ERR_BUSY until the oldest is collected, tickets collected once, blocks reserved up front, the
    calling thread's HIP device untouched."""
    if force_rccl:
        monkeypatch.setenv("F3DS_MULTI_FORCE_RCCL", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    prm = P.launch_params(voxel_res=0.012, seed_res=0.1)
    batches = [[P.synth_frame(0, 300 + 10 * b + i, 160 + 16 * i, 120, 30) for i in range(3 + b % 2)] for b in range(5)]
    want = [[oracle.segment(f, prm)[1] for f in fr] for fr in batches]
    mg = P.MultiGpu(n_devices=1, max_frames_per_device=4)
    mg.reserve(224 * 120)
    t0 = mg.submit(batches[0], prm); t1 = mg.submit(batches[1], prm)
    with pytest.raises(P.F3dsError) as e:
        mg.submit(batches[2], prm)
    assert e.value.code == P.ERR_BUSY
    got = {0: mg.collect(t0)[0]}
    t2 = mg.submit(batches[2], prm)
    got[1] = mg.collect(t1)[0]
    t3 = mg.submit(batches[3], prm)
    got[2] = mg.collect(t2)[0]; got[3] = mg.collect(t3)[0]
    got[4] = mg.segment(batches[4], prm)[0]                     # submit + collect in one
    assert mg.lib.f3ds_multi_collect(mg.handle, t3) == P.ERR_ARG     # a ticket is collected once
    for b in range(5):
        assert len(got[b]) == len(want[b]) and all(np.array_equal(g, w) for g, w in zip(got[b], want[b])), b
    mg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cap", ["0", "200"])
def test_relabel_as_two_kernels_matches_too(P, oracle, monkeypatch, cap):
    """Stage 6 is one kernel (d_relabel: union-find relabel into an LDS table, then the label write) while S0 + 1 <= 12288 and
    d_region_ids + d_point_labels beyond; F3DS_RELABEL_LDS_CAP (read per call) forces the second form on small frames, alone and
    in a batch that mixes frames below and above the cap."""
    monkeypatch.setenv("F3DS_RELABEL_LDS_CAP", cap)
    ctx = P.Context(0)
    for n in ["rgbd_320x240_ghosts", "fixture_launch_flags", "rgbd_160x120_threshold_1"]:
        pts, prm = case_points(P, n), case_params(P, n)
        lab = ctx.segment(pts, prm)
        assert sha_of(lab) == GOLD[n]["labels_sha256"], n
        rc, olab, ores, oh = oracle.segment(pts, prm)
        assert ctx.result.n_regions == ores.n_regions and same_bits(oh.get("VOXEL_REGION"), ctx.debug("VOXEL_REGION")), n
    ctx.close()
    prm = P.launch_params(voxel_res=0.012, seed_res=0.1)
    frames = [P.synth_frame(0, 70 + i, 160 + 80 * i, 120 + 60 * i, 30) for i in range(3)]      # ~60, ~170, ~330 supervoxels
    ctxs = [P.Context(0) for _ in frames]
    labels = P.segment_batch(ctxs, frames, prm)
    for f, l in zip(frames, labels):
        assert np.array_equal(l, oracle.segment(f, prm)[1])
    for c in ctxs:
        c.close()


@pytest.mark.gpu
def test_development_switches_are_ignored_without_f3ds_dev(P, oracle, monkeypatch):
    """VERDICT r5 item 6 on the device: with F3DS_DEV unset a stray F3DS_MERGE_NW / F3DS_MERGE_KEYS / F3DS_FORCE_GLOBAL_MERGE in the environment does not change
    which merge kernel a lone frame runs (F3DS_DBG_MERGE_LAYOUT); with the gate open the same variables do."""
    prm = P.launch_params(voxel_res=0.012, seed_res=0.1)
    pts = P.synth_frame(0, 4100, 200, 150, 30)
    want = oracle.segment(pts, prm)[1]
    ctx = P.Context(0)
    monkeypatch.delenv("F3DS_DEV", raising=False)
    monkeypatch.setenv("F3DS_MERGE_NW", "4"); monkeypatch.setenv("F3DS_MERGE_KEYS", "global")
    assert np.array_equal(ctx.segment(pts, prm), want)
    assert ctx.merge_layout() == (8, 2)
    monkeypatch.setenv("F3DS_FORCE_GLOBAL_MERGE", "1")
    assert np.array_equal(ctx.segment(pts, prm), want)
    assert ctx.merge_layout() == (8, 2)
    monkeypatch.setenv("F3DS_DEV", "1")
    assert np.array_equal(ctx.segment(pts, prm), want)
    assert ctx.merge_layout() == (0, 0)                  # d_merge
    monkeypatch.delenv("F3DS_FORCE_GLOBAL_MERGE")
    assert np.array_equal(ctx.segment(pts, prm), want)
    assert ctx.merge_layout() == (4, 0)
    ctx.close()


@pytest.mark.gpu
def test_merge_layout_follows_what_else_is_on_the_device(P, oracle):
    """choose_merge_kind: a lone frame and a lone batch call run the 8-wave merge loop; batch calls of 16 frames or more that overlap on one device
    take the 4-wave one.  Labels are the oracle's either way (F3DS_DBG_MERGE_LAYOUT is diagnostics only)."""
    import threading
    prm = P.launch_params(voxel_res=0.012, seed_res=0.1)
    frames = [P.synth_frame(0, 4100 + i, 200, 150, 30) for i in range(4)]
    want = [oracle.segment(f, prm)[1] for f in frames]
    ctx = P.Context(0)
    assert np.array_equal(ctx.segment(frames[0], prm), want[0])
    assert ctx.merge_layout() == (8, 2)
    groups = [[P.Context(0) for _ in range(16)] for _ in range(3)]
    labs = P.segment_batch(groups[0], [frames[i % 4] for i in range(16)], prm)
    assert all(np.array_equal(labs[i], want[i % 4]) for i in range(16))
    assert groups[0][0].merge_layout() == (8, 2)          # nothing else was running
    seen, errors = [], []

    def worker(g):
        try:
            for it in range(6):
                labs = P.segment_batch(groups[g], [frames[(i + g) % 4] for i in range(16)], prm)
                assert all(np.array_equal(labs[i], want[(i + g) % 4]) for i in range(16)), (g, it)
                seen.append(groups[g][0].merge_layout())
        except Exception as e:      # noqa
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(g,)) for g in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert set(seen) <= {(4, 0), (4, 2), (8, 2)} and ((4, 0) in seen or (4, 2) in seen), seen      # (a call that happens to reach its merge stage alone keeps 8 waves)
    for grp in groups:
        for c in grp:
            c.close()
    ctx.close()


@pytest.mark.gpu
def test_mixed_frame_sizes_on_concurrent_contexts_keep_their_state(P, oracle):
    """Scratch is sized by device-wide high-water marks.  A context must never lose frame state because ANOTHER
    context (another thread) met a larger frame: ENSURE only regrows a buffer that is too small for the request, and buffers
    below the mark are brought up to it at the start of a segment call.  Two threads feed their contexts frames 1.3x .. 3x
    apart in size, and between the frames use what carries state across calls -- recluster, refineSupervoxels, the automatic
    threshold, the getters -- against the oracle."""
    import threading
    prm = P.launch_params(voxel_res=0.012, seed_res=0.1)
    sizes = [(160, 120), (208, 156), (320, 240), (480, 360)]          # points: 1 : 1.7 : 4 : 9; consecutive ones 1.3x .. 3x apart per axis pair
    frames = [P.synth_frame(0, 900 + i, w, h, 30) for i, (w, h) in enumerate(sizes)]
    want = []
    for f in frames:
        rc, lab, res, oh = oracle.segment(f, prm)
        p2 = prm.copy(); p2.threshold = 0.12
        truth = (np.arange(len(f), dtype=np.uint32) // 977) % 7
        want.append(dict(labels=lab, recluster=oh.cluster(p2, len(f))[1], refine=oh.refine(2), auto=oh.auto_threshold(prm, truth, len(f), 0.1, 0.3, 0.05)))
    errors = []

    def worker(order):
        try:
            ctx = P.Context(0)
            for i in order:
                f, w = frames[i], want[i]
                assert np.array_equal(ctx.segment(f, prm), w["labels"]), ("segment", i)
                barrier.wait(timeout=120)                            # the other thread now starts a frame of another size
                p2 = prm.copy(); p2.threshold = 0.12
                assert np.array_equal(ctx.recluster(p2), w["recluster"]), ("recluster", i)
                got = ctx.refine_supervoxels(2)
                assert np.array_equal(got["voxel_label"], w["refine"]["voxel_label"]) and same_bits(got["voxel_normal"], w["refine"]["voxel_normal"]), ("refine", i)
                truth = (np.arange(len(f), dtype=np.uint32) // 977) % 7
                bt, bp, table, lab = ctx.auto_threshold(prm, truth, 0.1, 0.3, 0.05)
                assert bt == w["auto"][1] and np.array_equal(lab, w["auto"][4]), ("auto_threshold", i)
                assert np.array_equal(ctx.recluster(prm), w["labels"]), ("recluster back", i)
            ctx.close()
        except Exception as e:      # noqa
            errors.append(repr(e))
            try:
                barrier.abort()
            except Exception:
                pass

    barrier = threading.Barrier(2)
    ts = [threading.Thread(target=worker, args=(o,)) for o in ([0, 2, 1, 3], [3, 1, 2, 0])]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


@pytest.mark.gpu
def test_cli_gpus_unreadable_file_and_bench_flag(P, oracle, tmp_path):
    """--gpus: a PCD file that cannot be read fails the run (exit code 1, the file named on stderr) and is left out of the batch;
    the good files still get their labels.  --bench <frames>: synthetic frames through the pipelined multi-GPU driver, one JSON line."""
    import subprocess
    exe = os.path.join(ROOT, "fast-3d-pointcloud-segmentation_amd", "supervoxel_clustering")
    d = tmp_path / "in"; d.mkdir()
    good = P.synth_frame(0, 31, 160, 120, 30)
    P.write_pcd(str(d / "good.pcd"), good[:, :3], good[:, 3].copy().view(np.uint32))
    (d / "broken.pcd").write_bytes(b"# .PCD v0.7\nVERSION 0.7\nFIELDS x y z rgba\nSIZE 4 4 4 4\nTYPE F F F U\nCOUNT 1 1 1 1\nWIDTH 10\nHEIGHT 1\nPOINTS 10\nDATA binary\n\x00\x01")
    flags = ["-v", "0.02", "-s", "0.2", "--CVX", "--AL", "-t", "0.2"]
    r = subprocess.run([exe, "-d", str(d), "--gpus", "1", "--labels", str(tmp_path / "lab")] + flags, capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 1 and "broken.pcd" in r.stderr, r.stderr + r.stdout
    assert not os.path.exists(str(tmp_path / "lab.broken"))
    rc, olab, ores, _ = oracle.segment(P.read_pcd(str(d / "good.pcd")), P.launch_params(voxel_res=0.02, seed_res=0.2))
    assert np.array_equal(np.fromfile(str(tmp_path / "lab.good"), np.uint32), olab)
    r = subprocess.run([exe, "--bench", "20", "--gpus", "1", "-v", "0.008", "-s", "0.08", "--CVX", "--AL", "-t", "0.2"], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr + r.stdout
    line = json.loads(r.stdout.strip().splitlines()[-1])
    big = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_golden_big.json")))["config5_seed1000"]["summary"]
    assert line["frames"] == 20 and line["gpus"] == 1 and line["mpoints_per_s"] > 0
    r = subprocess.run([exe, "--bench", "4"], capture_output=True, text=True, cwd=str(tmp_path))      # no -t
    assert r.returncode == 1


@pytest.mark.gpu
@pytest.mark.parametrize("threads", ["256", "384"])
def test_voxel_normals_kernel_widths_agree_with_the_golden_normals(P, monkeypatch, threads):
    """d_normals_t<384> builds a tile's tables with six waves (lone frames, small calls), d_normals_t<256> with four (calls of 16 frames or more:
    a four-wave workgroup fits on a unit beside another call's merge loop).  Each forced here (F3DS_NORMALS_THREADS, read per call) on golden
    cases -- small noisy frames whose tiles overflow the one-ring / two-ring tables take the global-memory path inside the same launch -- and on the
    1M-point frame: normals, and what the sweeps make of the tile tables the kernel writes for them."""
    monkeypatch.setenv("F3DS_NORMALS_THREADS", threads)
    big = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_golden_big.json")))
    ctx = P.Context(0)
    for n in ["rgbd_320x240_ghosts", "rgbd_320x240_large_supervoxels", "fixture_launch_flags", "fused_200k_nan_lambda", "rgbd_160x120_rgb_metric"]:
        lab = ctx.segment(case_points(P, n), case_params(P, n))
        assert sha_of(lab) == GOLD[n]["labels_sha256"], n
        for w in ("VOXEL_NORMAL", "VOXEL_SVLABEL", "VOXEL_DIST", "SV_CENTROID"):
            assert sha_of(ctx.debug(w)) == GOLD[n]["sha256"][w], (n, w)
    e = big["config5_seed1061"]
    lab = ctx.segment(P.synth_frame(*e["synth"]), P.launch_params(**e["params"]))
    assert sha_of(lab) == e["labels_sha256"]
    tl = ctx.tile_list_lengths()              # F3DS_DBG_TILE_LIST_LEN: one entry per 128-voxel tile, a one-ring list of at most 448 voxels or "overflowed"
    assert len(tl) == (ctx.result.n_voxels + 127) // 128
    fits = tl[tl != 0xFFFFFFFF]
    assert len(fits) > 0.9 * len(tl) and fits.min() >= 1 and fits.max() <= 448
    for w in ("VOXEL_SVLABEL", "MERGES"):
        assert sha_of(ctx.debug(w)) == e["sha256"][w], w
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env", [dict(F3DS_SWEEP_TILES="0"), dict(F3DS_SWEEP_TILE_HOLES="3"), dict(F3DS_SWEEP_TILE_HOLES="2", F3DS_INC_SHIFT="32")],
                         ids=["global-gathers", "every-third-tile-global", "holes-and-always-incremental"])
def test_lds_tiled_sweeps_equal_the_global_gather_sweeps(P, env):
    """d_sweep_R_pre / d_sweep_claim read a voxel's 27 neighbours from an LDS copy of its 128-voxel tile's one-ring (tables written by
    d_normals); tiles whose one-ring overflowed, and sweeps with ghost leaves, gather from global memory instead.  The default (all
    tiles staged) is what every other test runs; here the global path alone, and a mix of both inside one launch (every n-th tile
    declared overflowed), with full and with incremental sweeps, against the golden hashes.  The switches are read per call except
    F3DS_INC_SHIFT (library load), hence the child process."""
    import subprocess, sys
    names = ["rgbd_320x240_ghosts", "rgbd_320x240_large_supervoxels", "fixture_launch_flags", "fused_200k_nan_lambda"]
    code = (
        "import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import conftest; from golden_cases import case_points, case_params\n"
        "P = conftest.pkg(); ctx = P.Context(0); out = {}\n"
        "for n in %r:\n"
        "    lab = ctx.segment(case_points(P, n), case_params(P, n))\n"
        "    out[n] = dict(labels=conftest.sha_of(lab), **{w: conftest.sha_of(ctx.debug(w)) for w in ('VOXEL_SVLABEL', 'VOXEL_DIST', 'SV_CENTROID', 'MERGES')})\n"
        "print(json.dumps(out))\n") % (ROOT, os.path.join(ROOT, "tests"), names)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    for n in names:
        assert got[n]["labels"] == GOLD[n]["labels_sha256"], (n, env)
        for w in ("VOXEL_SVLABEL", "VOXEL_DIST", "SV_CENTROID", "MERGES"):
            assert got[n][w] == GOLD[n]["sha256"][w], (n, w, env)


@pytest.mark.gpu
@pytest.mark.parametrize("G", [2, 3])
def test_multi_gpu_driver_logical_devices_run_the_g_gt_1_path(P, oracle, monkeypatch, G):
    """The G > 1 logic of f3ds_multi_* on a 1-GPU box (F3DS_MULTI_LOGICAL: `devices` names GPU 0 G times).  Every logical device has its own
    worker thread, contexts and label blocks; frame i runs on logical device i mod G; the per-device block / base / off arithmetic, the
    exchange thread with several devices and the two-slot pipeline all run -- only the wire differs (device-to-device copies where a real node
    uses ncclSend / ncclRecv: RCCL refuses duplicate GPUs).  Five ragged frames (one empty), pipelined submit / collect, labels == single
    calls, and the gathered block on devices[0] laid out [dev 0 | dev 1 | ...] in frame order."""
    monkeypatch.setenv("F3DS_MULTI_LOGICAL", "1")
    prm = P.launch_params(voxel_res=0.012, seed_res=0.1)
    batches = [[P.synth_frame(0, 700 + 10 * b + i, 120 + 24 * i, 90 + 10 * ((i + b) % 3), 30) for i in range(5)] for b in range(3)]
    batches[1][2] = np.zeros((0, 4), np.float32)                   # a device whose share holds an empty frame
    batches[2] = batches[2][:G - 1]                                # fewer frames than devices: the last device idles
    want = [[oracle.segment(f, prm)[1] for f in fr] for fr in batches]
    mg = P.MultiGpu(devices=[0] * G, max_frames_per_device=3)
    assert mg.devices() == G and [mg.device_of_frame(i) for i in range(5)] == [i % G for i in range(5)]
    t0 = mg.submit(batches[0], prm); t1 = mg.submit(batches[1], prm)
    with pytest.raises(P.F3dsError) as e:
        mg.submit(batches[2], prm)
    assert e.value.code == P.ERR_BUSY
    got0, res0 = mg.collect(t0)
    # the gathered block of batch 0 on devices[0]: device d's frames in frame order at the running offset of the devices before d
    hip = ctypes.CDLL("libamdhip64.so")
    blk = mg.lib.f3ds_multi_gathered_labels_of(mg.handle, t0)
    assert blk
    order = [i for d in range(G) for i in range(d, 5, G)]
    total = sum(len(batches[0][i]) for i in order)
    host = np.empty(total, np.uint32)
    assert hip.hipMemcpy(ctypes.c_void_p(host.ctypes.data), ctypes.c_void_p(blk), ctypes.c_size_t(total * 4), 2) == 0      # hipMemcpyDeviceToHost
    assert np.array_equal(host, np.concatenate([want[0][i] for i in order]))
    t2 = mg.submit(batches[2], prm)
    got1, _ = mg.collect(t1); got2, _ = mg.collect(t2)
    assert mg.lib.f3ds_multi_gathered_labels_of(mg.handle, t0) is None      # its slot now belongs to batch 2
    for b, got in enumerate((got0, got1, got2)):
        assert len(got) == len(want[b]) and all(np.array_equal(g, w) for g, w in zip(got, want[b])), b
    assert [r.n_points for r in res0] == [len(f) for f in batches[0]]
    with pytest.raises(P.F3dsError) as e:
        mg.segment(batches[0] * 2, prm)                            # 10 frames > G x 3 only when G = 2 or 3
    assert e.value.code == P.ERR_CAPACITY
    mg.close()
    monkeypatch.delenv("F3DS_MULTI_LOGICAL")
    with pytest.raises(P.F3dsError):
        P.MultiGpu(devices=[0, 0])                                 # duplicates stay an error outside the test mode


@pytest.mark.gpu
def test_bench_distributed_bookkeeping_with_one_forced_rank(tmp_path):
    """bench.py's N > 1 branch -- process group, per-step RCCL gather of the label block, barrier, max-over-ranks timing -- with one rank
    (F3DS_BENCH_FORCE_DIST=1) on small frames: the line is well-formed and says what it ran.  (A scaling curve needs a node: not claimed.)"""
    import json
    import subprocess
    import sys
    env = dict(os.environ, F3DS_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "64", "--groups", "2", "--width", "160", "--height", "120",
                        "--host-io-steps", "0", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode in (0, 1), r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["unit"] == "Mpoints/s" and line["scaling"] == "weak"
    assert "RCCL gather" in line["config"]["label_gather"] and line["config"]["frames_timed"] == 128
    assert line["roofline"]["kernel"].startswith("k_batched<d_merge") and line["library"].startswith("f3ds 1.2.0 src:")
    # 160x120 frames are not BASELINE's workload: the line carries no `value`, the rate sits under what_if_value
    assert line["value"] is None and "not the BASELINE workload" in line["invalid"] and line["what_if_value"] > 0


@pytest.mark.gpu
def test_bench_strong_mode_with_one_forced_rank(tmp_path):
    """bench.py --strong: BASELINE config 5 as written (ONE 64-frame batch per step for the whole job, frame i on rank i mod N, one gather per step), through
    the N > 1 branch with one forced rank.  N = 1 shards nothing, so this checks the bookkeeping (scaling field, seeds, label hashes at full size are the
    weak mode's); the two-rank sharding and reassembly run on gloo in tests/test_distributed_cpu.py.  UNMEASURED on N > 1 hardware."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, F3DS_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--strong", "--steps", "2", "--warmup", "1", "--batch", "64", "--groups", "2", "--width", "160",
                        "--height", "120", "--host-io-steps", "0", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode in (0, 1), r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["scaling"] == "strong" and line["config"]["frames_timed"] == 128 and line["config"]["frames_per_step_per_gpu"] == 64
    assert "ONE batch of 64" in line["config"]["workload"] and "RCCL gather" in line["config"]["label_gather"]
    est = line["strong_scaling_estimate"]
    assert est and est["n8_ms_estimated"] > 0 and "UNMEASURED" in est["what"]


@pytest.mark.gpu
def test_incident_list_pool_regrows_instead_of_failing(P, oracle, monkeypatch):
    """The merge loop leaves every merged region's incident-edge list in a pool (2 E entries for the initial lists + room for the lists that outgrow their
    segment).  F3DS_ILIST_SLACK=1 starts with a pool one merge in twenty would fit: the loop stops with the capacity flag, the host reruns the stage with four
    times the room (several times over), and the result is the oracle's -- as for the leaf pool and the weight-history arrays."""
    monkeypatch.setenv("F3DS_ILIST_SLACK", "1")
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_golden.json")))
    ctx = P.Context(0)
    for n in ["rgbd_320x240_ghosts", "fixture_launch_flags", "fused_200k_nan_lambda"]:
        lab = ctx.segment(case_points(P, n), case_params(P, n))
        assert sha_of(lab) == gold[n]["labels_sha256"], n
        assert sha_of(ctx.debug("MERGES")) == gold[n]["sha256"]["MERGES"], n
    ctxs = [P.Context(0) for _ in range(3)]      # a batch in which the frames need different pool sizes
    frames = [case_points(P, n) for n in ("rgbd_160x120", "rgbd_320x240_ghosts", "fixture_launch_flags")]
    prm = case_params(P, "rgbd_320x240_ghosts")
    labs = P.segment_batch(ctxs, frames, prm)
    for f, l in zip(frames, labs):
        assert np.array_equal(l, oracle.segment(f, prm)[1])
    for c in ctxs + [ctx]:
        c.close()


@pytest.mark.gpu
def test_pinned_host_label_buffers_are_written_by_the_kernel_itself(P, oracle, monkeypatch):
    """With F3DS_DIRECT_LABELS=1, host label buffers that are pinned (hipHostMalloc) get their labels straight from the relabel kernel through the buffer's
    device address; by default (and for pageable buffers) they take the staging copy on the device's copy stream, or on the call's own stream with
    F3DS_COPY_STREAM=0.  Same labels every way, for a batch with an empty frame and for a lone frame."""
    hip = ctypes.CDLL("libamdhip64.so")
    prm = P.launch_params(voxel_res=0.012, seed_res=0.1)
    frames = [P.synth_frame(0, 5200 + i, 200 + 20 * i, 150, 30) for i in range(3)] + [np.zeros((0, 4), np.float32)]
    want = [oracle.segment(f, prm)[1] for f in frames]
    bufs = []
    for f in frames:
        p = ctypes.c_void_p()
        assert hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(max(4 * len(f), 64)), 0) == 0
        bufs.append(p)
    ctxs = [P.Context(0) for _ in frames]
    try:
        for direct in ("direct", "copy stream", "own stream"):
            monkeypatch.setenv("F3DS_DIRECT_LABELS", "1" if direct == "direct" else "0")
            monkeypatch.setenv("F3DS_COPY_STREAM", "0" if direct == "own stream" else "1")
            for p, f in zip(bufs, frames):
                ctypes.memset(p, 0xAB, max(4 * len(f), 64))
            arrs = [np.ascontiguousarray(f, np.float32) for f in frames]
            P.segment_batch(ctxs, [a.ctypes.data for a in arrs], prm, labels_out=[p.value for p in bufs], n=[len(a) for a in arrs], raw_host=True)
            for p, f, w in zip(bufs, frames, want):
                got = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint32)), shape=(max(len(f), 1),))[:len(f)]
                assert np.array_equal(got, w), direct
        # a lone frame into the pinned buffer, direct path again
        monkeypatch.setenv("F3DS_DIRECT_LABELS", "1")
        a = np.ascontiguousarray(frames[1], np.float32)
        P.segment_batch(ctxs[:1], [a.ctypes.data], prm, labels_out=[bufs[1].value], n=[len(a)], raw_host=True)
        got = np.ctypeslib.as_array(ctypes.cast(bufs[1], ctypes.POINTER(ctypes.c_uint32)), shape=(len(a),))
        assert np.array_equal(got, want[1])
    finally:
        for c in ctxs:
            c.close()
        for p in bufs:
            hip.hipHostFree(p)


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["6", "60601"], ["6", "60602", "16"]], ids=["any-ratio", "big-supervoxels"])
def test_fuzz_mid_size_frames_against_the_oracle(args):
    """tools/fuzz_gpu_big.py under the driver (until round 6 this evidence was builder-run only): 6 seeded frames of 100k-350k points per mode (the
    oracle and its refineSupervoxels take 10-15 s per frame on the box; the builder runs hundreds) --
    any seed / voxel resolution ratio, and ratios >= 16 (supervoxels of hundreds of voxels: the merge loop's wide speculative merges, the wave-wide centroid path) --,
    random metrics / merging modes / thresholds / leaf orders, every intermediate array and refineSupervoxels against the oracle run on the box."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu_big.py")] + args, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert lines[-1] == "bad 0" and sum(" ok " in l for l in lines) == 6, r.stdout[-3000:]


@pytest.mark.gpu
def test_fuzz_degenerate_clouds_against_the_oracle():
    """tools/fuzz_clouds.py --gpu under the driver: 40 seeded degenerate clouds -- empty, one point, duplicates, lines, planes, NaN / inf,
    negative z, huge extents -- through the HIP path against the oracle, return codes and every intermediate array."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_clouds.py"), "40", "606", "--gpu"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "mismatches 0" in r.stdout.strip().splitlines()[-1], r.stdout[-2000:]
