"""Ground-truth evaluation and automatic threshold: the oracle's restatement of
Testing (reference src/testing.cpp) and all_thresh / best_thresh (src/clustering.cpp:691-774) against
hand-computed answers and an independent numpy formulation; GPU parity is in test_gpu_parity.py."""
import ctypes
import math

import numpy as np
import pytest

from golden_cases import case_params, case_points, synthetic_truth


def eval_clouds(oracle, P, sxyz, slab, txyz, tlab):
    sxyz = np.ascontiguousarray(sxyz, np.float32); txyz = np.ascontiguousarray(txyz, np.float32)
    slab = np.ascontiguousarray(slab, np.uint32); tlab = np.ascontiguousarray(tlab, np.uint32)
    out = P.Performance()
    rc = oracle.lib.f3ds_oracle_eval_clouds(ctypes.c_void_p(sxyz.ctypes.data), ctypes.c_void_p(slab.ctypes.data), ctypes.c_size_t(len(slab)),
                                            ctypes.c_void_p(txyz.ctypes.data), ctypes.c_void_p(tlab.ctypes.data), ctypes.c_size_t(len(tlab)),
                                            ctypes.byref(out))
    return rc, out


def numpy_scores(table, ssize, tsize, n):
    """Independent formulation of Testing's scores from a contingency table (float64, so only close)."""
    K, M = table.shape
    order = {}
    for j in range(M):
        order.setdefault(int(tsize[j]), j)               # std::map::insert keeps the first label of each size
    match = [-1] * M
    used = set()
    for size in sorted(order, reverse=True):
        j = order[size]
        col = table[:, j].astype(np.int64).copy()
        while True:
            row = int(np.argmax(col))
            if row not in used:
                break
            col[row] = 0
            if not col.any():
                row = -1
                break
        match[j] = row
        if row >= 0:
            used.add(row)
    p = r = fp = fn = w = 0.0
    for j in range(M):
        g = float(tsize[j])
        if match[j] < 0:
            fn += g
            continue
        i = match[j]
        inter, s = float(table[i, j]), float(ssize[i])
        p += inter * g / s; r += inter; fp += s - inter; fn += g - inter
        w += inter * g / (s + g - inter)
    hs = -sum(float(s) / n * math.log(float(s) / n) for s in ssize)
    ht = -sum(float(t) / n * math.log(float(t) / n) for t in tsize)
    mi = sum(float(table[i, j]) / n * math.log(n * float(table[i, j]) / (float(ssize[i]) * float(tsize[j])))
             for i in range(K) for j in range(M) if table[i, j])
    prec, rec = p / n, r / n
    return dict(voi=hs + ht - 2 * mi, precision=prec, recall=rec, fscore=0.0 if prec == 0 and rec == 0 else 2 * prec * rec / (prec + rec),
                wov=w / n, fpr=fp / n, fnr=fn / n)


def test_scores_hand_computed(P, oracle):
    """5 points, segments {0,0,0,1,1}, truth {0,0,1,1,1}: intersections [[2,1],[0,2]], matches 0->0, 1->1."""
    xyz = np.arange(15, dtype=np.float32).reshape(5, 3)
    rc, s = eval_clouds(oracle, P, xyz, [0, 0, 0, 1, 1], xyz, [0, 0, 1, 1, 1])
    assert rc == 0
    assert s.precision == pytest.approx((2 * 2 / 3 + 2 * 3 / 2) / 5, rel=1e-6)
    assert s.recall == pytest.approx(0.8, rel=1e-6) and s.fpr == pytest.approx(0.2, rel=1e-6) and s.fnr == pytest.approx(0.2, rel=1e-6)
    assert s.fscore == pytest.approx(2 * s.precision * s.recall / (s.precision + s.recall), rel=1e-6)
    assert s.wov == pytest.approx((2 * 2 / 3 + 2 * 3 / 3) / 5, rel=1e-6)
    h = lambda c: -sum(x / 5 * math.log(x / 5) for x in c)
    mi = sum(r / 5 * math.log(5 * r / (p * q)) for r, p, q in ((2, 3, 2), (1, 3, 3), (2, 2, 3)))
    assert s.voi == pytest.approx(h([3, 2]) + h([2, 3]) - 2 * mi, abs=1e-6)


def test_scores_identical_clouds_and_label_renumbering(P, oracle):
    xyz = np.random.default_rng(1).random((40, 3)).astype(np.float32)
    lab = np.repeat(np.arange(4, dtype=np.uint32), [4, 8, 12, 16])          # distinct sizes: every label is matched
    rc, s = eval_clouds(oracle, P, xyz, lab, xyz, lab * 10 + 3)             # label_map renumbers by ascending label
    assert rc == 0 and s.precision == 1.0 and s.recall == 1.0 and s.fscore == 1.0 and s.wov == 1.0 and s.fpr == 0.0 and s.fnr == 0.0
    assert abs(s.voi) < 1e-6


def test_scores_equal_size_truth_labels_share_one_match(P, oracle):
    """t_sizes is keyed by size (testing.cpp:97-100): of two truth labels of equal size only the first is matched."""
    xyz = np.arange(24, dtype=np.float32).reshape(8, 3)
    lab = np.repeat(np.arange(2, dtype=np.uint32), 4)
    rc, s = eval_clouds(oracle, P, xyz, lab, xyz, lab)
    assert rc == 0 and s.recall == 0.5 and s.fnr == 0.5 and s.precision == 0.5


def test_scores_disjoint_clouds(P, oracle):
    a = np.zeros((3, 3), np.float32); b = np.ones((3, 3), np.float32)
    rc, s = eval_clouds(oracle, P, a, [0, 0, 0], b, [0, 0, 0])
    assert rc == 0 and s.precision == 0 and s.recall == 0 and s.fscore == 0 and s.fnr == 1.0 and s.fpr == 1.0
    assert eval_clouds(oracle, P, a[:0], [], b, [0, 0, 0])[0] == P.ERR_ARG


def voxel_truth_labels(h, pts, truth, P):
    """Truth label of every voxel as main() builds it: mean label colour per voxel, numbered by first appearance."""
    pv = h.get("POINT_VOXEL")
    V = len(h.get("VOXEL_COUNT"))
    lut = np.array([P.label_color(i) for i in range(256)], np.uint32)[truth % 256]
    sums = np.zeros((V, 3), np.float64)
    ok = pv >= 0
    for k, sh in enumerate((16, 8, 0)):
        np.add.at(sums[:, k], pv[ok], ((lut[ok] >> sh) & 255).astype(np.float64))
    cnt = h.get("VOXEL_COUNT").astype(np.float32)
    mean = (sums.astype(np.float32) / cnt[:, None]).astype(np.uint32)
    col = (mean[:, 0] << 16) | (mean[:, 1] << 8) | mean[:, 2]
    ids, out = {}, np.zeros(V, np.uint32)
    for v, c in enumerate(col.tolist()):
        out[v] = ids.setdefault(c, len(ids))
    return out


@pytest.mark.parametrize("name", ["rgbd_160x120", "rgbd_320x240_ghosts", "fixture_launch_flags"])
def test_oracle_evaluate_against_numpy(P, oracle, name):
    pts = case_points(P, name)
    truth = synthetic_truth(pts) if name != "fixture_launch_flags" else np.zeros(len(pts), np.uint32)   # the fixture has no label field
    rc, labels, res, h = oracle.segment(pts, case_params(P, name))
    assert rc == 0
    rc, got = h.evaluate(truth)
    assert rc == 0
    tl = voxel_truth_labels(h, pts, truth, P)
    xyz, seg, _ = h.voxel_cloud()
    vx = h.get("VOXEL_XYZ").reshape(-1, 3)
    index = {tuple(r): i for i, r in enumerate(vx.tolist())}
    assert len(index) == len(vx)                                     # voxel centroids are distinct points
    K, M = int(seg.max()) + 1, int(tl.max()) + 1
    table = np.zeros((K, M), np.int64)
    seen = set()
    for r, s in zip(xyz.tolist(), seg.tolist()):
        v = index[tuple(r)]
        if (s, v) not in seen:                                       # a ghost leaf repeats a point inside one segment
            seen.add((s, v)); table[s, tl[v]] += 1
    want = numpy_scores(table, np.bincount(seg, minlength=K), np.bincount(tl, minlength=M), float(len(vx)))
    for k, v in want.items():
        assert getattr(got, k) == pytest.approx(v, rel=2e-4, abs=2e-5), k


def test_oracle_auto_threshold(P, oracle):
    pts = case_points(P, "rgbd_160x120")
    prm = case_params(P, "rgbd_160x120")
    truth = synthetic_truth(pts)
    rc, _, _, h = oracle.segment(pts, prm)
    assert rc == 0
    rc, bt, bp, table, labels = h.auto_threshold(prm, truth, len(pts), 0.05, 0.6, 0.05)
    assert rc == 0
    t, want = np.float32(0.05), []
    while t <= np.float32(0.6):
        want.append(float(t)); t = np.float32(t + np.float32(0.05))
    assert list(table) == want                                        # float accumulation of the reference's loop
    best = max(table.values(), key=lambda s: s["fscore"])["fscore"]
    first = next(t for t in table if table[t]["fscore"] == best)
    assert bt == first and bp.fscore == best
    p2 = P.Params(); ctypes.memmove(ctypes.byref(p2), ctypes.byref(prm), ctypes.sizeof(P.Params)); p2.threshold = bt
    rc, l2, _ = h.cluster(p2, len(pts))
    assert rc == 0 and np.array_equal(labels, l2)
    for t in (want[0], want[5]):                                      # each entry is the score of clustering at that threshold
        p2.threshold = t
        h.cluster(p2, len(pts))
        assert h.evaluate(truth)[1].as_dict() == table[t]
    # argument rules of all_thresh (clustering.cpp:693-705): out of range -> invalid_argument, swapped bounds accepted
    assert h.auto_threshold(prm, truth, len(pts), -0.1, 0.5, 0.1)[0] == P.ERR_OUT_OF_RANGE      # std::out_of_range, clustering.cpp:694-698
    assert h.auto_threshold(prm, truth, len(pts), 0.1, 1.5, 0.1)[0] == P.ERR_OUT_OF_RANGE
    rc, _, _, swapped, _ = h.auto_threshold(prm, truth, len(pts), 0.6, 0.05, 0.05)
    assert rc == 0 and swapped == table
