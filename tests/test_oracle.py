"""The oracle pinned against (a) the reference's own known-answer vectors, (b) facts measured on the
reference's bundled fixture, (c) the committed golden file."""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import sha_of, same_bits, ALL_DEBUG, FIXTURE_PCD, ROOT
from golden_cases import GOLDEN_CASES, case_params, case_points

KAT = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kat.json")))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_golden.json")))


def _f3(a):
    return (ctypes.c_float * 3)(*a)


@pytest.mark.parametrize("which", ["oracle", "emul"])
def test_ciede2000_table(oracle, emul, which):
    """34 rows of ColorUtilities::lab_test (src/color_utilities.cpp:354-460), table rounded to 4 decimals."""
    chk = oracle if which == "oracle" else emul
    f = chk.fn("ciede00"); f.restype = ctypes.c_float
    worst = 0.0
    for L1, a1, b1, L2, a2, b2, want in KAT["ciede2000"]:
        got = f(_f3([L1, a1, b1]), _f3([L2, a2, b2]))
        worst = max(worst, abs(got - want))
        assert abs(f(_f3([L2, a2, b2]), _f3([L1, a1, b1])) - want) < 1e-4      # symmetric to table precision
    assert worst < 1e-4


@pytest.mark.parametrize("which", ["oracle", "emul"])
def test_rgb_euclid_cases(oracle, emul, which):
    """7 cases of ColorUtilities::rgb_test (src/color_utilities.cpp:324-349)."""
    chk = oracle if which == "oracle" else emul
    f = chk.fn("rgb_eucl"); f.restype = ctypes.c_float
    for a, b, want in KAT["rgb_eucl"]:
        assert abs(f(_f3(a), _f3(b)) - want) < 1e-4
    assert abs(f(_f3([0, 0, 0]), _f3([255, 255, 255])) - KAT["RGB_RANGE"]) < 1e-4


def test_rgb2lab_sanity(oracle, emul):
    """convert_test (src/color_utilities.cpp:465-499) prints only; textbook values as sanity (cvtColor is unpinned)."""
    for chk in (oracle, emul):
        f = chk.fn("rgb2lab")
        lab = (ctypes.c_float * 3)()
        f(_f3([0, 0, 0]), lab); assert list(lab) == [0.0, 0.0, 0.0]
        f(_f3([255, 255, 255]), lab); assert abs(lab[0] - 100) < 0.01 and abs(lab[1]) < 0.01 and abs(lab[2]) < 0.01
        f(_f3([255, 255, 0]), lab); assert abs(lab[0] - 97.14) < 0.05 and abs(lab[1] + 21.55) < 0.1 and abs(lab[2] - 94.48) < 0.1
        f(_f3([123, 10, 200]), lab); assert 30 < lab[0] < 40 and lab[1] > 60 and lab[2] < -60
    # both restatements agree bit for bit
    rng = np.random.default_rng(5)
    for rgb in rng.uniform(0, 255, (500, 3)).astype(np.float32):
        l1 = (ctypes.c_float * 3)(); l2 = (ctypes.c_float * 3)()
        oracle.fn("rgb2lab")(_f3(rgb), l1); emul.fn("rgb2lab")(_f3(rgb), l2)
        assert bytes(l1) == bytes(l2)


def test_plane_normal_restatements_agree(oracle, emul):
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 5, 40, 757):
        for _ in range(20):
            base = rng.normal(0, 1, 3)
            pts = (base + rng.normal(0, 0.05, (n, 3)) * np.array([1, 1, 0.02])).astype(np.float32)
            vp = pts[0].copy()
            n1 = (ctypes.c_float * 4)(); n2 = (ctypes.c_float * 4)()
            oracle.fn("normal")(ctypes.c_void_p(pts.ctypes.data), ctypes.c_size_t(n), ctypes.c_void_p(vp.ctypes.data), n1)
            emul.fn("normal")(ctypes.c_void_p(pts.ctypes.data), ctypes.c_size_t(n), ctypes.c_void_p(vp.ctypes.data), n2)
            assert bytes(n1) == bytes(n2)
            if n >= 3:
                v = np.array(list(n1)[:3])
                assert abs(np.linalg.norm(v) - 1) < 1e-5 and np.dot(v, -vp) >= -1e-6     # unit, flipped to the origin


def test_fixture_facts(P, oracle):
    """SURVEY.md F8 / 8c cross-checks measured on the reference's bundled frame."""
    pts = P.read_pcd(FIXTURE_PCD)
    assert pts.shape == (307200, 4)
    fin = np.isfinite(pts[:, :3]).all(1)
    assert fin.sum() == 241407 and (pts[fin, 2] < 0).all()
    assert abs(pts[fin, 2].min() + 2.063) < 1e-3 and abs(pts[fin, 2].max() + 0.501) < 1e-3
    assert (pts[:, 3].view(np.uint32) >> 24 == 0).all()
    rc, labels, res, h = oracle.segment(pts, P.launch_params())
    assert rc == 0 and res.n_voxels == 34211 and res.octree_depth == 8 and res.sweeps == 16
    cnt = h.get("VOXEL_COUNT")
    assert cnt.max() == 25 and abs(cnt.mean() - 7.06) < 0.01
    g = h.get("GRID")
    assert g[3] == np.float32(0.008) and g[4] == 8
    assert (labels[~fin] == P.NO_LABEL).all()
    # default -v 0.008 -s 0.08: int(1.8f*s/v) = 17 -> 16 sweeps; -v 0.02 -s 0.2 -> 18 -> 17
    small = P.synth_frame(0, 7, 64, 48, 0)
    rc, _, r2, _ = oracle.segment(small, P.launch_params(voxel_res=0.02, seed_res=0.2))
    assert rc == 0 and r2.sweeps == 17


@pytest.mark.parametrize("name", list(GOLDEN_CASES))
@pytest.mark.parametrize("which", ["oracle", "emul"])
def test_golden(P, oracle, emul, name, which):
    """Oracle and device emulation both reproduce tests/golden/oracle_golden.json bit for bit."""
    chk = oracle if which == "oracle" else emul
    rc, labels, res, h = chk.segment(case_points(P, name), case_params(P, name))
    assert rc == 0
    g = GOLD[name]
    for k, v in g["summary"].items():
        got = getattr(res, k)
        assert (v == "nan" and got != got) or got == v, k
    for w in ALL_DEBUG:
        assert sha_of(h.get(w)) == g["sha256"][w], w
    assert sha_of(labels) == g["labels_sha256"]
    if "merges" in g:
        assert h.get("MERGES").reshape(-1, 3).tolist() == g["merges"]


def test_recluster_matches_fresh_run(P, oracle, emul):
    pts = case_points(P, "rgbd_160x120")
    for chk in (oracle, emul):
        rc, l0, r0, h = chk.segment(pts, case_params(P, "rgbd_160x120"))
        p2 = case_params(P, "rgbd_160x120_rgb_metric")
        rc2, l2, r2 = h.cluster(p2, len(pts))
        rc3, l3, r3, _ = chk.segment(pts, p2)
        assert rc == rc2 == rc3 == 0 and np.array_equal(l2, l3) and r2.n_merges == r3.n_merges


def test_libm_variant_distance(P, oracle, oracle_libm):
    """How far a libm-linked build (what a real PCL / OpenCV build would call) is from the shared-math build, measured on the
    reference's own fixture with the launch flags and asserted so that a change of csrc/f3ds_math.h that moves it is seen:
    voxel keys, seeds, supervoxel labels, adjacencies, the merge SEQUENCE (pairs) and the per-point labels are identical (label
    agreement 1.0); merge weights differ by <= 1.94e-7 (52 % of them in their last bit), voxel normals by <= 1.8e-7."""
    pts = P.read_pcd(FIXTURE_PCD)
    _, la, ra, ha = oracle.segment(pts, P.launch_params())
    _, lb, rb, hb = oracle_libm.segment(pts, P.launch_params())
    assert oracle_libm.lib.f3ds_oracle_uses_libm() == 1 and oracle.lib.f3ds_oracle_uses_libm() == 0
    assert ra.n_voxels == rb.n_voxels == 34211 and ra.n_supervoxels == rb.n_supervoxels and ra.n_regions == rb.n_regions and ra.n_merges == rb.n_merges
    for w in ("VOXEL_KEYS", "SEED_KEPT", "VOXEL_SVLABEL", "EDGES"):
        assert np.array_equal(ha.get(w), hb.get(w)), w
    ma, mb = ha.get("MERGES").reshape(-1, 3), hb.get("MERGES").reshape(-1, 3)
    assert np.array_equal(ma[:, :2], mb[:, :2])                                    # the same merges in the same order
    wa, wb = ma[:, 2].copy().view(np.float32), mb[:, 2].copy().view(np.float32)
    assert np.abs(wa - wb).max() <= 4e-7                                           # tolerance: two float ulps at weight ~0.2 (measured 1.94e-7)
    na, nb = ha.get("VOXEL_NORMAL"), hb.get("VOXEL_NORMAL")
    fin = np.isfinite(na) & np.isfinite(nb)
    assert np.abs(na[fin] - nb[fin]).max() <= 4e-7                                 # (measured 1.8e-7)
    assert float(np.mean(la == lb)) == 1.0                                         # measured label agreement on the fixture


def test_edge_cases(P, oracle, emul):
    prm = P.launch_params(voxel_res=0.02, seed_res=0.2)
    nan = np.float32("nan")
    cases = {
        "empty": np.zeros((0, 4), np.float32),
        "all_nan": np.full((10, 4), nan, np.float32),
        "single": np.array([[0.1, 0.2, 1.0, 0]], np.float32),
        "two_identical": np.array([[0.1, 0.2, 1.0, 0], [0.1, 0.2, 1.0, 0]], np.float32),
        "z_zero_and_inf": np.array([[0.1, 0.2, 0.0, 0], [0.3, 0.1, 1.0, 0], [0.3, 0.1, np.inf, 0], [0.5, 0.5, 2.0, 0]], np.float32),
        "negative_z_folded": np.array([[0.1, 0.2, -1.0, 0], [0.1, 0.2, 1.0, 0]], np.float32),
    }
    for name, pts in cases.items():
        ro = oracle.segment(pts, prm); re = emul.segment(pts, prm)
        assert ro[0] == 0 and re[0] == 0, name
        assert np.array_equal(ro[1], re[1]), name
        assert ro[2].n_voxels == re[2].n_voxels and ro[2].n_regions == re[2].n_regions, name
    o = oracle.segment(cases["negative_z_folded"], prm)
    assert o[2].n_voxels == 1


def test_refine_supervoxels_properties(P, oracle):
    """refineSupervoxels restatement (row N3, [PCL-recall]): zero iterations hand back the extract state; after real
    iterations the survivors are a subset of the supervoxels, every voxel they hold is counted once, none of them is
    empty, owned normals are unit vectors, the result is a pure function of the extract state, and the oracle's main
    state (supervoxels, clustering) is untouched."""
    pts = case_points(P, "rgbd_160x120"); prm = case_params(P, "rgbd_160x120")
    rc, lab, res, oh = oracle.segment(pts, prm)
    assert rc == 0
    sv0 = oh.get("VOXEL_SVLABEL").copy(); labels0 = oh.get("SV_LABELS").copy(); cent0 = oh.get("SV_CENTROID").reshape(-1, 10).copy()
    r0 = oh.refine(0)
    assert np.array_equal(r0["voxel_label"], sv0) and np.array_equal(r0["label"], labels0)
    assert np.array_equal(r0["xyz"].view(np.uint32), cent0[:, 0:3].view(np.uint32)) and np.array_equal(r0["normal"].view(np.uint32), cent0[:, 6:9].view(np.uint32))
    r3 = oh.refine(3)
    assert set(r3["label"]) <= set(labels0) and len(r3["label"]) > 0.8 * len(labels0)
    owned = r3["voxel_label"] != 0
    counts = np.bincount(r3["voxel_label"][owned], minlength=int(labels0.max()) + 1)
    for l, n in zip(r3["label"], r3["n_voxels"]):
        assert n >= counts[l] >= 1 and n - counts[l] <= 1          # a leaf held without being owned adds at most one
    assert counts.sum() == owned.sum() and set(np.nonzero(counts)[0]) == set(r3["label"])
    nn = np.linalg.norm(r3["voxel_normal"][owned], axis=1)
    assert np.all((np.abs(nn - 1) < 1e-5) | (nn == 0))
    assert (r3["voxel_label"] != sv0).any()                         # it did move boundaries
    r3b = oh.refine(3)
    for k in r3:
        assert np.array_equal(r3[k].view(np.uint32), r3b[k].view(np.uint32)), k      # a pure function of the extract state
    assert np.array_equal(oh.get("VOXEL_SVLABEL"), sv0) and np.array_equal(oh.get("SV_CENTROID").reshape(-1, 10).view(np.uint32), cent0.view(np.uint32))
    rc2, lab2, _ = oh.cluster(prm, len(pts))
    assert rc2 == 0 and np.array_equal(lab2, lab)


@pytest.mark.parametrize("name", ["rgbd_160x120", "rgbd_320x240_ghosts", "fixture_launch_flags"])
def test_refine_device_formulation_matches_oracle_on_cpu(P, oracle, emul, name):
    """Row N3 without a GPU: the device's way of refining (tests/emul: last-writer rule for the normals of ghost leaves,
    brute-force reseed, the R-predicate sweeps with kept centroids and seedless helpers) against the literal restatement."""
    pts = case_points(P, name); prm = case_params(P, name)
    rc, olab, ores, oh = oracle.segment(pts, prm)
    rc2, elab, eres, eh = emul.segment(pts, prm)
    assert rc == 0 and rc2 == 0 and np.array_equal(olab, elab)
    sv0 = eh.get("VOXEL_SVLABEL").copy()
    for k in (1, 3):
        want = oh.refine(k); got = eh.refine(k)
        for key in want:
            assert want[key].shape == got[key].shape and same_bits(want[key], got[key]), (k, key)
    assert np.array_equal(eh.get("VOXEL_SVLABEL"), sv0)
    rc3, lab3, _ = eh.cluster(prm, len(pts))
    assert rc3 == 0 and np.array_equal(lab3, olab)


def test_refine_device_formulation_random_frames_cpu(P, oracle, emul):
    rng = np.random.default_rng(2024)
    for i in range(10):
        w, h = int(rng.integers(30, 110)), int(rng.integers(24, 80))
        pts = P.synth_frame(0, 8100 + i, w, h, int(rng.integers(0, 100)))
        vres = float(rng.choice([0.02, 0.03, 0.05]))
        prm = P.launch_params(voxel_res=vres, seed_res=vres * float(rng.choice([2, 3, 5, 10])), use_transform=int(rng.integers(0, 2)), leaf_order=int(rng.integers(0, 2)))
        rc, olab, ores, oh = oracle.segment(pts, prm)
        rc2, elab, eres, eh = emul.segment(pts, prm)
        assert rc == rc2
        if rc:
            continue
        k = int(rng.integers(1, 4))
        want = oh.refine(k); got = eh.refine(k)
        for key in want:
            assert want[key].shape == got[key].shape and same_bits(want[key], got[key]), (i, k, key)
