"""BASELINE.json's full sizes.  The 1M frame is still compared with the oracle (a few seconds of
CPU); the 20M scene is checked through size-independent properties."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import sha_of, ALL_DEBUG, ROOT, first_mismatch

pytestmark = pytest.mark.gpu
BIG = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_golden_big.json")))      # tools/make_golden_big.py
SUMMARY = ("n_points", "n_finite", "n_voxels", "octree_depth", "n_seed_cells", "n_seeds", "n_supervoxels", "n_edges", "n_merges", "n_regions", "sweeps")


def _sha(a):
    return sha_of(a)


def _invariants(P, pts, labels, res):
    fin = np.isfinite(pts[:, :3]).all(1)
    assert (labels[~fin] == P.NO_LABEL).all()
    lab = labels[labels != P.NO_LABEL]
    assert lab.max() == res.n_regions - 1 and len(np.unique(lab)) == res.n_regions     # ids are dense 0..K-1
    assert res.n_regions == res.n_supervoxels - res.n_merges
    assert res.n_voxels <= res.n_finite and res.n_seeds <= res.n_seed_cells


@pytest.mark.parametrize("stage0", ["sort", "tiles"])
def test_1m_frame_matches_oracle(P, oracle, gpu_ctx, stage0, monkeypatch):
    monkeypatch.setenv("F3DS_VOX_TILES", "2" if stage0 == "tiles" else "0")       # stage 0 by sorting the points (a lone frame's path) | by tiles (the frames of a batch)
    pts = P.synth_frame(0, 1000, 1000, 1000, 30)
    prm = P.launch_params()
    labels = gpu_ctx.segment(pts, prm)
    res = gpu_ctx.result
    _invariants(P, pts, labels, res)
    rc, olab, ores, oh = oracle.segment(pts, prm)
    assert rc == 0 and np.array_equal(labels, olab)
    problems = [m for m in (first_mismatch(w, oh.get(w), gpu_ctx.debug(w)) for w in ALL_DEBUG) if m]      # all 20 intermediate arrays
    assert not problems, "\n".join(problems)
    gold = BIG["config5_seed1000"]
    assert _sha(labels) == gold["labels_sha256"] and _sha(gpu_ctx.debug("MERGES")) == gold["sha256"]["MERGES"]
    again = gpu_ctx.segment(pts, prm)                         # determinism: run twice, bit compare
    assert np.array_equal(labels, again)


@pytest.mark.parametrize("stage0", ["sort", "tiles"])
def test_nyu_scale_frame_matches_oracle(P, oracle, gpu_ctx, stage0, monkeypatch):
    monkeypatch.setenv("F3DS_VOX_TILES", "2" if stage0 == "tiles" else "0")
    pts = P.synth_frame(0, 2000, 640, 480, 200)
    prm = P.launch_params()
    labels = gpu_ctx.segment(pts, prm)
    _invariants(P, pts, labels, gpu_ctx.result)
    rc, olab, ores, oh = oracle.segment(pts, prm)
    assert rc == 0 and np.array_equal(labels, olab)
    problems = [m for m in (first_mismatch(w, oh.get(w), gpu_ctx.debug(w)) for w in ALL_DEBUG) if m]      # all 20 intermediate arrays
    assert not problems, "\n".join(problems)


def test_organised_5m_frame_on_stage0s_tile_path(P, monkeypatch):
    """Stage 0's tile path beyond the bench frame's size: an ORGANISED 2 560 x 2 048 frame (5.2M points, 137 k voxels, -v 0.006 -s 0.06) through
    the tile path and through the sort path -- stage-0 arrays, normals, supervoxel labels, merges and per-point labels against the oracle's committed hashes
    (tools/make_golden_big.py, ~40 s of CPU in the build container).  The 20M scene below is unorganised and always falls back to the sort path."""
    gold = BIG["organised_5m_frame"]
    pts = P.synth_frame(*gold["synth"])
    prm = P.launch_params(**gold["params"])
    ctx = P.Context(0)
    for mode, want_path in (("2", "tiles"), ("0", "sort")):
        monkeypatch.setenv("F3DS_VOX_TILES", mode)
        labels = ctx.segment(pts, prm)
        assert ctx.stage0_path() == want_path, mode
        assert {k: getattr(ctx.result, k) for k in SUMMARY} == gold["summary"], mode
        assert _sha(labels) == gold["labels_sha256"], mode
        for w, h in gold["sha256"].items():
            assert _sha(ctx.debug(w)) == h, (mode, w)
    _invariants(P, pts, labels, ctx.result)
    ctx.close()


def test_20m_scene_properties(P, gpu_ctx):
    pts = P.synth_frame(1, 3000, 5000, 4000, 0)
    prm = P.launch_params(voxel_res=0.02, seed_res=0.2, use_transform=0)
    labels = gpu_ctx.segment(pts, prm)
    res = gpu_ctx.result
    r1 = {k: getattr(res, k) for k in ("n_voxels", "n_seeds", "n_supervoxels", "n_edges", "n_merges", "n_regions")}
    _invariants(P, pts, labels, res)
    # pinned by the oracle's hashes of the same scene (generated in the build container, ~50 s of CPU there)
    gold = BIG["config4_20m_scene"]
    assert {k: getattr(res, k) for k in SUMMARY} == gold["summary"]
    assert _sha(labels) == gold["labels_sha256"]
    for w, h in gold["sha256"].items():
        assert _sha(gpu_ctx.debug(w)) == h, w
    assert res.sweeps == 17 and res.n_finite == len(pts)
    # points of one voxel share a label; every point sits in the voxel its coordinates say
    pv = gpu_ctx.debug("POINT_VOXEL"); vr = gpu_ctx.debug("VOXEL_REGION")
    assert np.array_equal(labels, vr[pv])
    keys = gpu_ctx.debug("VOXEL_KEYS").reshape(-1, 3); grid = gpu_ctx.debug("GRID")
    sample = np.random.default_rng(0).integers(0, len(pts), 200000)
    k = ((pts[sample, :3].astype(np.float64) - grid[:3]) / grid[3]).astype(np.uint32)
    assert np.array_equal(k, keys[pv[sample]])
    # leaf order = ascending Morton code of the keys
    def morton(kk):
        c = np.zeros(len(kk), np.uint64)
        for b in range(int(grid[4]) - 1, -1, -1):
            c = (c << np.uint64(3)) | (((kk[:, 0] >> b) & 1).astype(np.uint64) << np.uint64(2)) | (((kk[:, 1] >> b) & 1).astype(np.uint64) << np.uint64(1)) | ((kk[:, 2] >> b) & 1).astype(np.uint64)
        return c
    m = morton(keys)
    assert (m[1:] > m[:-1]).all()
    # voxel centroids are means of their points (float32 order-dependent sums -> small tolerance)
    cnt = gpu_ctx.debug("VOXEL_COUNT"); xyz = gpu_ctx.debug("VOXEL_XYZ").reshape(-1, 3)
    assert cnt.sum() == len(pts)
    sums = np.zeros((len(cnt), 3)); np.add.at(sums, pv, pts[:, :3].astype(np.float64))
    assert np.abs(sums / cnt[:, None] - xyz).max() < 1e-4      # tolerance: f32 sequential sums of <= ~200 values of magnitude <= 8
    # idempotence / determinism: the same frame again gives the same bits
    again = gpu_ctx.segment(pts, prm)
    assert np.array_equal(labels, again)
    assert r1 == {k: getattr(gpu_ctx.result, k) for k in r1}
    # merge stopping rule: every recorded merge weight is below the threshold and non-decreasing ties aside
    mg = gpu_ctx.debug("MERGES").reshape(-1, 3)
    w = mg[:, 2].copy().view(np.float32)
    assert (w < prm.threshold).all() and (mg[:, 0] < mg[:, 1]).all()


def test_config5_batch_of_64_1m_frames_matches_oracle_hashes(P):
    """BASELINE.json config 5 on one GPU: the 64 distinct 1M-point frames (seeds 1000..1063) as ONE f3ds_segment_batch call,
    device buffers in and out like bench.py; labels, merge sequence, supervoxel labels and counts of every frame against
    the oracle's committed hashes."""
    import torch
    dev = torch.device("cuda", 0)
    prm = P.launch_params(voxel_res=0.008, seed_res=0.08)
    seeds = list(range(1000, 1064))
    npts = 1000 * 1000
    ctxs = [P.Context(0) for _ in seeds]
    frames = [torch.from_numpy(P.synth_frame(0, s, 1000, 1000, 30)).to(dev) for s in seeds]
    block = torch.empty((len(seeds), npts), dtype=torch.int32, device=dev)
    P.segment_batch(ctxs, [f.data_ptr() for f in frames], prm, labels_out=[block[i].data_ptr() for i in range(len(seeds))], n=[npts] * len(seeds), on_device=True)
    torch.cuda.synchronize()
    labels = block.cpu().numpy().view(np.uint32)
    for i, s in enumerate(seeds):
        gold = BIG["config5_seed%d" % s]
        assert {k: getattr(ctxs[i].result, k) for k in SUMMARY} == gold["summary"], s
        assert _sha(labels[i]) == gold["labels_sha256"], s
        for w, h in gold["sha256"].items():
            assert _sha(ctxs[i].debug(w)) == h, (s, w)
    # device-mode frame, then Clustering::cluster again on it (the label array is sized from the frame)
    p2 = prm.copy(); p2.threshold = 0.1
    again = ctxs[3].recluster(p2)
    assert len(again) == npts and ctxs[3].result.n_regions > BIG["config5_seed1003"]["summary"]["n_regions"]
    assert _sha(ctxs[3].recluster(prm)) == BIG["config5_seed1003"]["labels_sha256"]
    for c in ctxs:
        c.close()


def test_step_pipeline_with_rccl_gather_one_rank(P):
    """bench.py's N > 1 loop with the real backend (torch "nccl" = RCCL), world size 1: batch calls on host threads through
    libf3ds, one gather per 4-frame step from the gather thread; rank 0's gathered blocks hold the labels of single calls."""
    import importlib
    import torch
    import torch.distributed as dist
    B = importlib.import_module("fast-3d-pointcloud-segmentation_amd.batch")
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()      # a free port: nothing else on the box may hold a fixed one
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        FPS, NB, W, H = 4, 4, 320, 240
        npts = W * H
        prm = P.launch_params(voxel_res=0.012, seed_res=0.1)
        host = [P.synth_frame(0, 50 + i, W, H, 40) for i in range(FPS)]
        frames = [torch.from_numpy(f).to(dev) for f in host]
        ctxs = [[P.Context(0) for _ in range(6)] for _ in range(2)]
        blocks = [torch.zeros((FPS, npts), dtype=torch.int32, device=dev) for _ in range(NB)]
        bufs = [torch.empty((FPS, npts), dtype=torch.int32, device=dev)]
        side = torch.cuda.Stream(device=dev)
        got = []

        def run_batch(g, f0, f1):
            k = f1 - f0
            P.segment_batch(ctxs[g][:k], [frames[f % FPS].data_ptr() for f in range(f0, f1)], prm,
                            labels_out=[blocks[(f // FPS) % NB][f % FPS].data_ptr() for f in range(f0, f1)], n=[npts] * k, on_device=True)

        def on_step(s):
            with torch.cuda.stream(side):
                out = B.gather_label_block(blocks[s % NB], dist, bufs, dst=0)
            side.synchronize()
            got.append(out[0].cpu().numpy().view(np.uint32).copy())

        B.StepPipeline(FPS, 6, 2, NB, run_batch, on_step).run(5)
        single = P.Context(0)
        want = np.stack([single.segment(h, prm) for h in host])
        assert len(got) == 5 and all(np.array_equal(g, want) for g in got)
        single.close()
        for grp in ctxs:
            for c in grp:
                c.close()
    finally:
        dist.destroy_process_group()
