import ctypes
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

try:                      # torch bundles its own HIP runtime: when both live in one process torch must load first,
    import torch  # noqa   # libf3ds then binds to the runtime that is already mapped
except ImportError:
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("F3DS_DEV", "1")      # the tests drive the library's development switches (kernel layouts, test hooks): csrc/f3ds_dev.h reads them only behind this gate
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_stamp_checked = False


def pkg():
    """The product package (its directory name has hyphens, so import it by string).  On first use the loaded libf3ds.so is checked
    against the sources on disk (build stamp, f3ds_version_string): a stale prebuilt library fails every test that touches it."""
    global _stamp_checked
    P = importlib.import_module("fast-3d-pointcloud-segmentation_amd")
    if not _stamp_checked:
        P.check_library_is_current()
        _stamp_checked = True
    return P


def _make(directory, target=None):
    cmd = ["make", "-s", "-C", os.path.join(ROOT, directory)] + ([target] if target else [])
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)


class CpuChecker:
    """ctypes wrapper shared by the oracle (prefix f3ds_oracle) and the device emulation (f3ds_emul)."""

    def __init__(self, path, prefix):
        self.lib = ctypes.CDLL(path)
        self.prefix = prefix
        self.P = pkg()

    def fn(self, name):
        return getattr(self.lib, self.prefix + "_" + name)

    def segment(self, pts, params):
        P = self.P
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 4)
        labels = np.empty(len(pts), np.uint32)
        res = P.Result()
        h = ctypes.c_void_p()
        rc = self.fn("segment")(ctypes.c_void_p(pts.ctypes.data), ctypes.c_size_t(len(pts)), ctypes.byref(params),
                                ctypes.c_void_p(labels.ctypes.data), ctypes.byref(res), ctypes.byref(h))
        return rc, labels, res, Handle(self, h)

    def cluster_supervoxels(self, sv, pairs, params):
        """f3ds_oracle_cluster_supervoxels: set_initialstate(segm, adj) + cluster(threshold) on caller-supplied supervoxels (arrays as
        Context.cluster_supervoxels takes them).  Returns (rc, region_of_sv, voxel_labels, result, handle)."""
        P = self.P
        a = {k: np.ascontiguousarray(sv[k], np.float32 if k in ("voxel_xyz", "centroid_xyz", "normal") else np.uint32) for k in
             ("label", "voxel_offset", "voxel_xyz", "voxel_rgba", "centroid_xyz", "normal")}
        S = len(a["label"])
        st = P.SupervoxelSet(S, *[a[k].ctypes.data for k in ("label", "voxel_offset", "voxel_xyz", "voxel_rgba", "centroid_xyz", "normal")])
        pairs = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
        nvox = int(a["voxel_offset"][S]) if S else 0
        region = np.zeros(S, np.uint32); vlab = np.zeros(nvox, np.uint32)
        res = P.Result(); h = ctypes.c_void_p()
        rc = self.fn("cluster_supervoxels")(ctypes.byref(st), ctypes.c_void_p(pairs.ctypes.data), ctypes.c_size_t(len(pairs)), ctypes.byref(params),
                                            ctypes.c_void_p(region.ctypes.data), ctypes.c_void_p(vlab.ctypes.data), ctypes.byref(res), ctypes.byref(h))
        return rc, region, vlab, res, Handle(self, h)


class Handle:
    def __init__(self, chk, h):
        self.chk, self.h = chk, h

    def get(self, name):
        P = self.chk.P
        nb = ctypes.c_size_t()
        assert self.chk.fn("get")(self.h, P.DBG[name], None, ctypes.c_size_t(0), ctypes.byref(nb)) == 0
        buf = np.zeros(nb.value, np.uint8)
        assert self.chk.fn("get")(self.h, P.DBG[name], ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(nb.value), ctypes.byref(nb)) == 0
        return buf.view(P.DBG_DTYPE[name])

    def cluster(self, params, n):
        P = self.chk.P
        labels = np.empty(n, np.uint32)
        res = P.Result()
        rc = self.chk.fn("cluster")(self.h, ctypes.byref(params), ctypes.c_void_p(labels.ctypes.data), ctypes.byref(res))
        return rc, labels, res

    def evaluate(self, truth):
        P = self.chk.P
        t = np.ascontiguousarray(truth, np.uint32)
        out = P.Performance()
        rc = self.chk.fn("evaluate")(self.h, ctypes.c_void_p(t.ctypes.data), ctypes.byref(out))
        return rc, out

    def auto_threshold(self, params, truth, n, start=0.8, end=1.0, step=0.005):
        P = self.chk.P
        t = np.ascontiguousarray(truth, np.uint32)
        cap = 4096
        ts = np.zeros(cap, np.float32); ps = (P.Performance * cap)()
        cnt = ctypes.c_size_t(); bt = ctypes.c_float(); bp = P.Performance()
        labels = np.empty(n, np.uint32)
        rc = self.chk.fn("auto_threshold")(self.h, ctypes.byref(params), ctypes.c_void_p(t.ctypes.data), ctypes.c_float(start), ctypes.c_float(end),
                                           ctypes.c_float(step), ctypes.c_void_p(ts.ctypes.data), ps, ctypes.c_size_t(cap), ctypes.byref(cnt),
                                           ctypes.byref(bt), ctypes.byref(bp), ctypes.c_void_p(labels.ctypes.data))
        table = {float(ts[i]): ps[i].as_dict() for i in range(min(cnt.value, cap))}
        return rc, bt.value, bp, table, labels

    def refine(self, num_itr):
        V = len(self.get("VOXEL_SVLABEL")); S = len(self.get("SV_LABELS"))
        vl = np.zeros(V, np.uint32); vn = np.zeros((V, 3), np.float32)
        lab = np.zeros(S, np.uint32); feat = np.zeros((S, 10), np.float32); cnt = np.zeros(S, np.uint32); k = ctypes.c_size_t()
        fn = self.chk.fn("refine")
        fn.restype = ctypes.c_int
        rc = fn(self.h, ctypes.c_int(num_itr), ctypes.c_void_p(vl.ctypes.data), ctypes.c_void_p(vn.ctypes.data), ctypes.c_void_p(lab.ctypes.data),
                ctypes.c_void_p(feat.ctypes.data), ctypes.c_void_p(cnt.ctypes.data), ctypes.c_size_t(S), ctypes.byref(k))
        assert rc == 0
        k = k.value
        return dict(voxel_label=vl, voxel_normal=vn, label=lab[:k], xyz=feat[:k, 0:3], rgb=feat[:k, 3:6], normal=feat[:k, 6:9], n_voxels=cnt[:k])

    def export_supervoxels(self):
        """The supervoxel_clusters map + getSupervoxelAdjacency multimap of the frame, in the array form f3ds_cluster_supervoxels takes
        (what main() hands to set_initialstate, supervoxel_clustering.cpp:424).  Returns (dict of arrays incl. voxel_leaf, pairs)."""
        fn = self.chk.fn("export_supervoxels")
        ns, nv, npairs = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        z = ctypes.c_size_t(0)
        assert fn(self.h, None, None, None, None, None, None, None, z, z, ctypes.byref(ns), ctypes.byref(nv), None, z, ctypes.byref(npairs)) == 0
        S, V, E = ns.value, nv.value, npairs.value
        sv = dict(label=np.zeros(S, np.uint32), voxel_offset=np.zeros(S + 1, np.uint32), voxel_xyz=np.zeros((V, 3), np.float32), voxel_rgba=np.zeros(V, np.uint32),
                  voxel_leaf=np.zeros(V, np.uint32), centroid_xyz=np.zeros((S, 3), np.float32), normal=np.zeros((S, 3), np.float32))
        pairs = np.zeros((E, 2), np.uint32)
        vp = lambda a: ctypes.c_void_p(a.ctypes.data)
        assert fn(self.h, vp(sv["label"]), vp(sv["voxel_offset"]), vp(sv["voxel_xyz"]), vp(sv["voxel_rgba"]), vp(sv["voxel_leaf"]), vp(sv["centroid_xyz"]), vp(sv["normal"]),
                  ctypes.c_size_t(S), ctypes.c_size_t(V), ctypes.byref(ns), ctypes.byref(nv), vp(pairs), ctypes.c_size_t(E), ctypes.byref(npairs)) == 0
        return sv, pairs

    def regions(self):
        """get_currentstate().first as Context.regions() returns it."""
        fn = self.chk.fn("regions")
        n = ctypes.c_size_t()
        assert fn(self.h, None, None, None, None, None, ctypes.c_size_t(0), ctypes.byref(n)) == 0
        k = n.value
        out = dict(label=np.zeros(k, np.uint32), n_voxels=np.zeros(k, np.uint32), xyz=np.zeros((k, 3), np.float32), normal=np.zeros((k, 3), np.float32),
                   rgb=np.zeros((k, 3), np.float32))
        vp = lambda a: ctypes.c_void_p(a.ctypes.data)
        assert fn(self.h, vp(out["label"]), vp(out["n_voxels"]), vp(out["xyz"]), vp(out["normal"]), vp(out["rgb"]), ctypes.c_size_t(k), ctypes.byref(n)) == 0
        return out

    def region_voxels(self):
        fn = self.chk.fn("region_voxels")
        n = ctypes.c_size_t()
        assert fn(self.h, None, None, None, ctypes.c_size_t(0), ctypes.byref(n)) == 0
        xyz = np.zeros((n.value, 3), np.float32); rgba = np.zeros(n.value, np.uint32); idx = np.zeros(n.value, np.uint32)
        assert fn(self.h, ctypes.c_void_p(xyz.ctypes.data), ctypes.c_void_p(rgba.ctypes.data), ctypes.c_void_p(idx.ctypes.data), ctypes.c_size_t(n.value), ctypes.byref(n)) == 0
        return xyz, rgba, idx

    def voxel_cloud(self):
        n = ctypes.c_size_t()
        self.chk.fn("voxel_cloud")(self.h, None, None, None, ctypes.c_size_t(0), ctypes.byref(n))
        xyz = np.zeros((n.value, 3), np.float32); lab = np.zeros(n.value, np.uint32); rgba = np.zeros(n.value, np.uint32)
        rc = self.chk.fn("voxel_cloud")(self.h, ctypes.c_void_p(xyz.ctypes.data), ctypes.c_void_p(lab.ctypes.data), ctypes.c_void_p(rgba.ctypes.data),
                                        ctypes.c_size_t(n.value), ctypes.byref(n))
        assert rc == 0
        return xyz, lab, rgba

    def close(self):
        if self.h:
            self.chk.fn("free")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@pytest.fixture(scope="session")
def P():
    return pkg()


@pytest.fixture(scope="session")
def oracle():
    p = os.path.join(ROOT, "oracle", "libf3ds_oracle.so")
    _make("oracle")           # (a no-op when the library is newer than its sources)
    return CpuChecker(p, "f3ds_oracle")


@pytest.fixture(scope="session")
def oracle_libm():
    p = os.path.join(ROOT, "oracle", "libf3ds_oracle_libm.so")
    _make("oracle")
    return CpuChecker(p, "f3ds_oracle")


@pytest.fixture(scope="session")
def emul():
    p = os.path.join(ROOT, "tests", "emul", "libf3ds_emul.so")
    _make("tests/emul")
    return CpuChecker(p, "f3ds_emul")


@pytest.fixture(scope="session")
def gpu_ctx(P):
    if P.device_count() < 1:
        pytest.fail("GPU test selected but libf3ds sees no HIP device (no CPU fallback exists)")
    ctx = P.Context(0)
    yield ctx
    ctx.close()


def same_bits(a, b):
    """Byte equality, except that a NaN matches a NaN: the payload / sign of a NaN produced by 0/0 differs between the
    host (x86: 0xFFC00000) and the device (0x7FC00000) and carries no information (float arrays only)."""
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.dtype.kind != "f":
        return a.tobytes() == b.tobytes()
    an, bn = np.isnan(a), np.isnan(b)
    return bool(np.array_equal(an, bn)) and a[~an].tobytes() == b[~bn].tobytes()


FIXTURE_PCD = os.path.join(ROOT, "tests", "golden", "milk_cartoon_all_small_clorox.pcd")

ALL_DEBUG = ["GRID", "VOXEL_KEYS", "VOXEL_COUNT", "VOXEL_XYZ", "VOXEL_RGB", "VOXEL_NORMAL", "VOXEL_NEIGHBORS", "POINT_VOXEL", "SEED_ORIG",
             "SEED_KEPT", "VOXEL_SVLABEL", "VOXEL_DIST", "SV_LABELS", "SV_CENTROID", "EDGES", "EDGE_DELTAS", "EDGE_WEIGHTS", "MERGES",
             "VOXEL_REGION", "SV_REGION"]


def canon(a):
    """Bytes of an array with every NaN replaced by the canonical quiet NaN: a NaN's sign / payload depends on the compiler's
    operand order (x86 propagates the first operand's) and on the platform (0/0 is 0xFFC00000 on the host, 0x7FC00000 on the
    device) and carries no information.  Integer arrays pass unchanged."""
    a = np.ascontiguousarray(a)
    if a.dtype.kind == "f":
        a = a.copy()
        a[np.isnan(a)] = np.nan
    return a


def sha_of(a):
    import hashlib
    return hashlib.sha256(canon(a).tobytes()).hexdigest()


def bits_equal(a, b):
    """Bit-for-bit equality (NaN payloads included)."""
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def first_mismatch(name, a, b):
    if a.shape != b.shape:
        return "%s: shape %s vs %s" % (name, a.shape, b.shape)
    a, b = canon(a), canon(b)
    d = np.nonzero(a.view(np.uint8).reshape(-1) != b.view(np.uint8).reshape(-1))[0]
    if len(d) == 0:
        return None
    i = d[0] // a.dtype.itemsize
    return "%s: %d differing bytes, first at element %d: %r vs %r" % (name, len(d), i, a.reshape(-1)[i], b.reshape(-1)[i])
