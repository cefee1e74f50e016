"""f3ds -- MI355X-native supervoxel + hierarchical-merge segmenter (host-side binding).

Thin ctypes layer over the C-ABI in ``include/f3ds.h`` (``libf3ds.so``, hand-written HIP for
gfx950).  The classes mirror the reference's interface for this path:

* :class:`Clustering`   -- ``/root/reference/include/supervoxel_clustering/clustering.h:84-212``
  (same setter names, same exceptions: ``logic_error`` -> :class:`LogicError`,
  ``invalid_argument`` -> ``ValueError``), fed by :class:`SupervoxelClustering`, the stand-in
  for ``pcl::SupervoxelClustering<PointXYZRGBA>`` as ``main()`` drives it
  (``/root/reference/src/supervoxel_clustering.cpp:348-367``).
* :func:`segment`       -- the whole frame in one call (what ``main()`` does between ``:313``
  and ``:449``).

There is no CPU fallback: constructing a :class:`Context` without a GPU, or with the shared
library missing, raises.  PyTorch is not required; device pointers (e.g. ``tensor.data_ptr()``)
can be passed with ``on_device=True``.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
def _dev_env(name):
    """A development switch of the environment: only read while F3DS_DEV is set (csrc/f3ds_dev.h; the library applies the same gate)."""
    return os.environ.get(name) if os.environ.get("F3DS_DEV", "0") not in ("", "0") else None


LIB_PATH = _dev_env("F3DS_LIB") or os.path.join(_HERE, "libf3ds.so")       # F3DS_LIB (with F3DS_DEV=1): A/B builds during development

NO_LABEL = 0xFFFFFFFF
LAB_CIEDE00, RGB_EUCL = 0, 1
NORMALS_DIFF, CONVEX_NORMALS_DIFF = 0, 1
MANUAL_LAMBDA, ADAPTIVE_LAMBDA, EQUALIZATION = 0, 1, 2

# debug selectors (include/f3ds.h)
DBG = dict(GRID=0, VOXEL_KEYS=1, VOXEL_COUNT=2, VOXEL_XYZ=3, VOXEL_RGB=4, VOXEL_NORMAL=5, VOXEL_NEIGHBORS=6,
           POINT_VOXEL=7, SEED_ORIG=8, SEED_KEPT=9, VOXEL_SVLABEL=10, VOXEL_DIST=11, SV_LABELS=12, SV_CENTROID=13,
           EDGES=14, EDGE_DELTAS=15, EDGE_WEIGHTS=16, MERGES=17, VOXEL_REGION=18, SV_REGION=19)
DBG_DTYPE = dict(GRID=np.float64, VOXEL_KEYS=np.uint32, VOXEL_COUNT=np.uint32, VOXEL_XYZ=np.float32, VOXEL_RGB=np.float32,
                 VOXEL_NORMAL=np.float32, VOXEL_NEIGHBORS=np.int32, POINT_VOXEL=np.int32, SEED_ORIG=np.int32, SEED_KEPT=np.int32,
                 VOXEL_SVLABEL=np.uint32, VOXEL_DIST=np.float32, SV_LABELS=np.uint32, SV_CENTROID=np.float32, EDGES=np.uint32,
                 EDGE_DELTAS=np.float32, EDGE_WEIGHTS=np.float32, MERGES=np.uint32, VOXEL_REGION=np.uint32, SV_REGION=np.uint32)


class F3dsError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("f3ds error %d: %s" % (code, text))
        self.code = code


class LogicError(F3dsError):
    """std::logic_error of the reference (clustering.cpp:576,591,671)."""


class Params(ctypes.Structure):
    _fields_ = [("voxel_res", ctypes.c_float), ("seed_res", ctypes.c_float), ("w_color", ctypes.c_float),
                ("w_spatial", ctypes.c_float), ("w_normal", ctypes.c_float), ("use_transform", ctypes.c_int32),
                ("color_metric", ctypes.c_int32), ("geom_metric", ctypes.c_int32), ("merging", ctypes.c_int32),
                ("lambda_", ctypes.c_float), ("bins", ctypes.c_int32), ("threshold", ctypes.c_float),
                ("leaf_order", ctypes.c_int32), ("fold_negative_z", ctypes.c_int32)]

    def copy(self):
        p = Params()
        ctypes.memmove(ctypes.byref(p), ctypes.byref(self), ctypes.sizeof(Params))
        return p


class Result(ctypes.Structure):
    _fields_ = [("n_points", ctypes.c_uint64), ("n_finite", ctypes.c_uint64), ("n_voxels", ctypes.c_uint32),
                ("octree_depth", ctypes.c_uint32), ("n_seed_cells", ctypes.c_uint32), ("n_seeds", ctypes.c_uint32),
                ("n_supervoxels", ctypes.c_uint32), ("n_edges", ctypes.c_uint32), ("n_merges", ctypes.c_uint32),
                ("n_regions", ctypes.c_uint32), ("sweeps", ctypes.c_uint32), ("lambda_", ctypes.c_float),
                ("ms_total", ctypes.c_float), ("ms_stage", ctypes.c_float * 8)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "ms_stage"}
        d["ms_stage"] = list(self.ms_stage)
        return d


class SupervoxelSet(ctypes.Structure):
    """f3ds_supervoxel_set (include/f3ds.h): the supervoxel_clusters map of the reference as plain arrays."""
    _fields_ = [("n_supervoxels", ctypes.c_uint32), ("label", ctypes.c_void_p), ("voxel_offset", ctypes.c_void_p), ("voxel_xyz", ctypes.c_void_p),
                ("voxel_rgba", ctypes.c_void_p), ("centroid_xyz", ctypes.c_void_p), ("normal", ctypes.c_void_p)]


_lib = None


class Performance(ctypes.Structure):
    """performanceSet (include/supervoxel_clustering/testing.h:40-48)."""
    _fields_ = [(k, ctypes.c_float) for k in ("voi", "precision", "recall", "fscore", "wov", "fpr", "fnr")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def load_library(path=None):
    """Load libf3ds.so (built by ``__graft_entry__.build()`` / ``make -C csrc``).  Raises if missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ImportError("libf3ds.so not found at %s -- build it first (python -c 'import __graft_entry__ as g; g.build()'); "
                          "this package has no CPU fallback" % p)
    lib = ctypes.CDLL(p)
    vp, sz, u32p = ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint32)
    lib.f3ds_default_params.argtypes = [ctypes.POINTER(Params)]; lib.f3ds_default_params.restype = None
    lib.f3ds_version.restype = ctypes.c_int
    if hasattr(lib, "f3ds_version_string"):      # (F3DS_LIB may point at an older build during A/B runs)
        lib.f3ds_version_string.restype = ctypes.c_char_p
    lib.f3ds_strerror.argtypes = [ctypes.c_int]; lib.f3ds_strerror.restype = ctypes.c_char_p
    lib.f3ds_last_hip_error.restype = ctypes.c_char_p
    lib.f3ds_device_count.restype = ctypes.c_int
    lib.f3ds_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]; lib.f3ds_create.restype = ctypes.c_int
    lib.f3ds_destroy.argtypes = [vp]; lib.f3ds_destroy.restype = None
    lib.f3ds_set_stream.argtypes = [vp, vp]; lib.f3ds_set_stream.restype = ctypes.c_int
    lib.f3ds_segment.argtypes = [vp, vp, sz, ctypes.c_int, ctypes.POINTER(Params), vp, ctypes.c_int, ctypes.POINTER(Result)]
    lib.f3ds_segment.restype = ctypes.c_int
    lib.f3ds_segment_batch.argtypes = [ctypes.POINTER(vp), ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(sz), ctypes.c_int, ctypes.POINTER(Params),
                                       ctypes.POINTER(vp), ctypes.c_int, ctypes.POINTER(Result)]
    lib.f3ds_segment_batch.restype = ctypes.c_int
    lib.f3ds_recluster.argtypes = [vp, ctypes.POINTER(Params), vp, ctypes.c_int, ctypes.POINTER(Result)]
    lib.f3ds_recluster.restype = ctypes.c_int
    lib.f3ds_evaluate.argtypes = [vp, vp, ctypes.POINTER(Performance)]; lib.f3ds_evaluate.restype = ctypes.c_int
    lib.f3ds_auto_threshold.argtypes = [vp, ctypes.POINTER(Params), vp, ctypes.c_float, ctypes.c_float, ctypes.c_float, vp, vp, sz, ctypes.POINTER(sz),
                                        ctypes.POINTER(ctypes.c_float), ctypes.POINTER(Performance), vp, ctypes.c_int, ctypes.POINTER(Result)]
    lib.f3ds_auto_threshold.restype = ctypes.c_int
    lib.f3ds_get_voxel_centroid_cloud.argtypes = [vp, vp, vp, vp, sz, ctypes.POINTER(sz)]; lib.f3ds_get_voxel_centroid_cloud.restype = ctypes.c_int
    lib.f3ds_get_supervoxels.argtypes = [vp, vp, vp, vp, vp, vp, sz, ctypes.POINTER(sz)]; lib.f3ds_get_supervoxels.restype = ctypes.c_int
    lib.f3ds_get_supervoxel_adjacency.argtypes = [vp, vp, sz, ctypes.POINTER(sz)]; lib.f3ds_get_supervoxel_adjacency.restype = ctypes.c_int
    lib.f3ds_refine_supervoxels.argtypes = [vp, ctypes.c_int]; lib.f3ds_refine_supervoxels.restype = ctypes.c_int
    lib.f3ds_get_refined_voxels.argtypes = [vp, vp, vp, sz, ctypes.POINTER(sz)]; lib.f3ds_get_refined_voxels.restype = ctypes.c_int
    lib.f3ds_get_refined_supervoxels.argtypes = [vp, vp, vp, vp, vp, vp, sz, ctypes.POINTER(sz)]; lib.f3ds_get_refined_supervoxels.restype = ctypes.c_int
    lib.f3ds_get_voxel_cloud.argtypes = [vp, vp, vp, vp, sz, ctypes.POINTER(sz)]; lib.f3ds_get_voxel_cloud.restype = ctypes.c_int
    lib.f3ds_get_debug.argtypes = [vp, ctypes.c_int, vp, sz, ctypes.POINTER(sz)]; lib.f3ds_get_debug.restype = ctypes.c_int
    lib.f3ds_get_region_adjacency.argtypes = [vp, vp, sz, ctypes.POINTER(sz)]; lib.f3ds_get_region_adjacency.restype = ctypes.c_int
    lib.f3ds_cluster_supervoxels.argtypes = [vp, ctypes.POINTER(SupervoxelSet), vp, sz, ctypes.POINTER(Params), vp, vp, ctypes.POINTER(Result)]
    lib.f3ds_cluster_supervoxels.restype = ctypes.c_int
    lib.f3ds_get_regions.argtypes = [vp, vp, vp, vp, vp, vp, sz, ctypes.POINTER(sz)]; lib.f3ds_get_regions.restype = ctypes.c_int
    lib.f3ds_get_region_voxels.argtypes = [vp, vp, vp, vp, sz, ctypes.POINTER(sz)]; lib.f3ds_get_region_voxels.restype = ctypes.c_int
    lib.f3ds_multi_create.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)]; lib.f3ds_multi_create.restype = ctypes.c_int
    lib.f3ds_multi_destroy.argtypes = [vp]; lib.f3ds_multi_destroy.restype = None
    lib.f3ds_multi_devices.argtypes = [vp]; lib.f3ds_multi_devices.restype = ctypes.c_int
    lib.f3ds_multi_device_of_frame.argtypes = [vp, ctypes.c_int]; lib.f3ds_multi_device_of_frame.restype = ctypes.c_int
    lib.f3ds_multi_segment.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(sz), ctypes.c_int, ctypes.POINTER(Params), ctypes.POINTER(vp), ctypes.POINTER(Result)]
    lib.f3ds_multi_segment.restype = ctypes.c_int
    lib.f3ds_multi_submit.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(sz), ctypes.c_int, ctypes.POINTER(Params), ctypes.POINTER(vp), ctypes.POINTER(Result), ctypes.POINTER(ctypes.c_int)]
    lib.f3ds_multi_submit.restype = ctypes.c_int
    lib.f3ds_multi_collect.argtypes = [vp, ctypes.c_int]; lib.f3ds_multi_collect.restype = ctypes.c_int
    lib.f3ds_multi_reserve.argtypes = [vp, sz]; lib.f3ds_multi_reserve.restype = ctypes.c_int
    lib.f3ds_multi_gathered_labels.argtypes = [vp]; lib.f3ds_multi_gathered_labels.restype = vp
    if hasattr(lib, "f3ds_multi_gathered_labels_of"):
        lib.f3ds_multi_gathered_labels_of.argtypes = [vp, ctypes.c_int]; lib.f3ds_multi_gathered_labels_of.restype = vp
    lib.f3ds_multi_last_error.restype = ctypes.c_char_p
    lib.f3ds_stream_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)]; lib.f3ds_stream_create.restype = ctypes.c_int
    lib.f3ds_stream_destroy.argtypes = [vp]; lib.f3ds_stream_destroy.restype = None
    lib.f3ds_stream_buffer.argtypes = [vp, sz, ctypes.POINTER(vp)]; lib.f3ds_stream_buffer.restype = ctypes.c_int
    lib.f3ds_stream_submit.argtypes = [vp, vp, sz, ctypes.POINTER(Params), ctypes.c_uint64]; lib.f3ds_stream_submit.restype = ctypes.c_int
    lib.f3ds_stream_next.argtypes = [vp, vp, sz, ctypes.POINTER(sz), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(Result), ctypes.c_int]
    lib.f3ds_stream_next.restype = ctypes.c_int
    lib.f3ds_stream_peek.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(sz), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(Result), ctypes.c_int]
    lib.f3ds_stream_peek.restype = ctypes.c_int
    lib.f3ds_stream_pending.argtypes = [vp]; lib.f3ds_stream_pending.restype = ctypes.c_int
    lib.f3ds_pcd_read.argtypes = [ctypes.c_char_p, vp, vp, sz, ctypes.POINTER(sz), u32p, u32p]; lib.f3ds_pcd_read.restype = ctypes.c_int
    lib.f3ds_pcd_write.argtypes = [ctypes.c_char_p, vp, vp, vp, sz, ctypes.c_int]; lib.f3ds_pcd_write.restype = ctypes.c_int
    lib.f3ds_synth_frame.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, vp]
    lib.f3ds_synth_frame.restype = ctypes.c_int
    lib.f3ds_label_color.argtypes = [ctypes.c_uint32]; lib.f3ds_label_color.restype = ctypes.c_uint32
    if path is None:
        _lib = lib
    return lib


def source_stamp():
    """First 16 hex digits of the SHA-256 over the library's sources as csrc/Makefile hashes them (every *.hip *.inc *.h *.cpp of
    csrc/ but the generated stamp header, in byte-wise name order, then include/f3ds.h and include/f3ds_clustering.hpp)."""
    import glob
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    names = sorted(os.path.basename(f) for pat in ("*.hip", "*.inc", "*.h", "*.cpp") for f in glob.glob(os.path.join(csrc, pat)))
    files = [os.path.join(csrc, n) for n in names if n != "f3ds_build_stamp.h"]
    inc = os.path.join(os.path.dirname(_HERE), "include")
    files += [os.path.join(inc, "f3ds.h"), os.path.join(inc, "f3ds_clustering.hpp")]
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def library_stamp(lib=None):
    """The stamp the loaded libf3ds.so was built from (f3ds_version_string: 'f3ds 1.2.0 src:<stamp>[ +whatif]')."""
    text = (lib or load_library()).f3ds_version_string().decode()
    return text.split("src:")[1].split()[0], text


def merge_layout_info(n_edges, waves=8, keys_in_lds=2):
    """f3ds_merge_layout_info: (dynamic LDS bytes, rows a speculative second merge may absorb, edge slots, fits?) of d_merge_il_t<waves, keys_in_lds>
    for a frame with n_edges adjacencies.  Host arithmetic only."""
    out = (ctypes.c_uint32 * 4)()
    lib = load_library()
    _check(lib, lib.f3ds_merge_layout_info(ctypes.c_uint32(int(n_edges)), int(waves), int(keys_in_lds), out))
    return int(out[0]), int(out[1]), int(out[2]), bool(out[3])


def check_library_is_current(lib=None):
    """Raise if the loaded libf3ds.so was not built from the sources beside it (a stale prebuilt .so travels to the GPU box with the
    snapshot; measuring or testing it would describe some other code).  Skipped when F3DS_LIB points at another build on purpose."""
    if _dev_env("F3DS_LIB"):
        return
    have, text = library_stamp(lib)
    want = source_stamp()
    if have != want:
        raise ImportError("libf3ds.so is stale: built from sources %s, the tree holds %s (%s) -- rebuild with make -C %s" % (have, want, text, os.path.join(_HERE, "csrc")))


(OK, ERR_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_DEPTH, ERR_LOGIC, ERR_RANGE, ERR_UNSUPPORTED, ERR_IO, ERR_EQ_BIN,
 ERR_CAPACITY, ERR_BUSY, ERR_EMPTY, ERR_OUT_OF_RANGE) = (0, -1, -2, -3, -4, -5, -6, -7, -8, -9, -10, -11, -12, -13)       # include/f3ds.h:37-56


def _check(lib, rc):
    if rc == 0:
        return
    text = lib.f3ds_strerror(rc).decode()
    if rc == -3:
        text += " [" + lib.f3ds_last_hip_error().decode() + "]"
    if rc == -5:
        raise LogicError(rc, text)
    if rc == -6:
        raise ValueError(text)
    if rc == -13:
        raise IndexError(text)      # std::out_of_range of the reference (map::at, all_thresh bounds)
    raise F3dsError(rc, text)


def pack_supervoxels(segm):
    """{label: dict(voxels_xyz (n,3) f32, voxels_rgba (n,) u32, centroid (3,), normal (3,))} -- the shape of the reference's
    ``std::map<uint32_t, pcl::Supervoxel::Ptr>`` -- -> dict of the arrays f3ds_supervoxel_set points at (rows in the dict's own order)."""
    labels = np.array(list(segm.keys()), np.uint32)
    counts = [len(np.asarray(segm[k]["voxels_xyz"]).reshape(-1, 3)) for k in segm]
    off = np.zeros(len(labels) + 1, np.uint32)
    off[1:] = np.cumsum(counts)
    xyz = np.concatenate([np.asarray(segm[k]["voxels_xyz"], np.float32).reshape(-1, 3) for k in segm]) if len(labels) else np.zeros((0, 3), np.float32)
    rgba = np.concatenate([np.asarray(segm[k]["voxels_rgba"], np.uint32).reshape(-1) for k in segm]) if len(labels) else np.zeros(0, np.uint32)
    cent = np.array([np.asarray(segm[k]["centroid"], np.float32)[:3] for k in segm], np.float32).reshape(-1, 3)
    nrm = np.array([np.asarray(segm[k]["normal"], np.float32)[:3] for k in segm], np.float32).reshape(-1, 3)
    return dict(label=labels, voxel_offset=off, voxel_xyz=np.ascontiguousarray(xyz), voxel_rgba=np.ascontiguousarray(rgba), centroid_xyz=cent, normal=nrm)


def default_params(**kw):
    lib = load_library()
    p = Params()
    lib.f3ds_default_params(ctypes.byref(p))
    for k, v in kw.items():
        if k == "lambda":
            k = "lambda_"
        if not hasattr(p, k):
            raise TypeError("unknown parameter %r" % k)
        setattr(p, k, v)
    return p


def launch_params(**kw):
    """The reference's canonical flags ``--CVX --AL -t 0.2`` (launch/supervoxel_clustering.launch:3-6)."""
    d = dict(geom_metric=CONVEX_NORMALS_DIFF, merging=ADAPTIVE_LAMBDA, threshold=0.2)
    d.update(kw)
    return default_params(**d)


def device_count():
    return load_library().f3ds_device_count()


# ---- host helpers either side of the path -------------------------------------------------------
def read_pcd(path, with_labels=False):
    """PCD v0.7 (ascii / binary / binary_compressed) -> (N,4) float32 view of {x,y,z,rgba-bits}."""
    lib = load_library()
    n = ctypes.c_size_t(); w = ctypes.c_uint32(); h = ctypes.c_uint32()
    _check(lib, lib.f3ds_pcd_read(path.encode(), None, None, 0, ctypes.byref(n), ctypes.byref(w), ctypes.byref(h)))
    pts = np.zeros((n.value, 4), np.float32)
    labels = np.zeros(n.value, np.uint32) if with_labels else None
    _check(lib, lib.f3ds_pcd_read(path.encode(), pts.ctypes.data, labels.ctypes.data if with_labels else None, n.value,
                                  ctypes.byref(n), None, None))
    return (pts, labels) if with_labels else pts


def write_pcd(path, xyz, rgba=None, labels=None, binary=True):
    lib = load_library()
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    n = len(xyz)
    rgba = None if rgba is None else np.ascontiguousarray(rgba, np.uint32)
    labels = None if labels is None else np.ascontiguousarray(labels, np.uint32)
    _check(lib, lib.f3ds_pcd_write(path.encode(), xyz.ctypes.data, None if rgba is None else rgba.ctypes.data,
                                   None if labels is None else labels.ctypes.data, n, 1 if binary else 0))


def synth_frame(kind, seed, width, height, nan_permille=0):
    """Deterministic synthetic frame; (N,4) float32, column 3 = rgba bits."""
    lib = load_library()
    pts = np.zeros((width * height, 4), np.float32)
    _check(lib, lib.f3ds_synth_frame(kind, seed, width, height, nan_permille, pts.ctypes.data))
    return pts


def label_color(label):
    return load_library().f3ds_label_color(label)


# ---- device context ------------------------------------------------------------------------------
class Context:
    """One (device, stream) pair with its grow-only scratch.  Not thread-safe; one per GPU/stream."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = ctypes.c_void_p()
        _check(self.lib, self.lib.f3ds_create(device, ctypes.byref(h)))
        self.handle = h
        self.device = device
        self.result = Result()
        self._n = 0

    def close(self):
        if getattr(self, "handle", None):
            self.lib.f3ds_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream_ptr):
        _check(self.lib, self.lib.f3ds_set_stream(self.handle, ctypes.c_void_p(hip_stream_ptr)))

    def segment(self, points, params, labels_out=None, n=None, on_device=False):
        """points: (N,4) float32 ndarray (host) or a device pointer (int) with ``n`` given and
        ``on_device=True``.  Returns the per-point labels (ndarray; ``labels_out`` itself when a host array is given) or writes
        them to the device pointer ``labels_out``."""
        if on_device:
            ptr, count = ctypes.c_void_p(int(points)), int(n)
            out_ptr = ctypes.c_void_p(int(labels_out)) if labels_out is not None else None
            _check(self.lib, self.lib.f3ds_segment(self.handle, ptr, count, 1, ctypes.byref(params), out_ptr, 1, ctypes.byref(self.result)))
            self._n = count
            return None
        pts = np.ascontiguousarray(points, np.float32).reshape(-1, 4)
        if labels_out is None:
            labels = np.empty(len(pts), np.uint32)
        else:       # the caller's own buffer (reused from call to call: a fresh 80 MB array for a 20M-point scene is 20 000 page faults inside the download)
            labels = labels_out
            if not (isinstance(labels, np.ndarray) and labels.dtype == np.uint32 and labels.flags.c_contiguous and labels.size == len(pts)):
                raise ValueError("labels_out must be a contiguous uint32 array with one entry per point")
        _check(self.lib, self.lib.f3ds_segment(self.handle, pts.ctypes.data, len(pts), 0, ctypes.byref(params), labels.ctypes.data, 0,
                                               ctypes.byref(self.result)))
        self._n = len(pts)
        return labels

    def recluster(self, params):
        labels = np.empty(int(self.result.n_points) or self._n, np.uint32)      # f3ds_recluster writes one label per point of the frame
        _check(self.lib, self.lib.f3ds_recluster(self.handle, ctypes.byref(params), labels.ctypes.data, 0, ctypes.byref(self.result)))
        return labels

    def evaluate(self, truth_point_labels):
        """Scores of the current segmentation against per-point ground-truth labels (Testing::eval_performance)."""
        t = np.ascontiguousarray(truth_point_labels, np.uint32)
        if len(t) != self._n:
            raise ValueError("one ground-truth label per input point is required")
        out = Performance()
        _check(self.lib, self.lib.f3ds_evaluate(self.handle, t.ctypes.data, ctypes.byref(out)))
        return out

    def auto_threshold(self, params, truth_point_labels, start=0.8, end=1.0, step=0.005):
        """all_thresh + best_thresh; returns (best threshold, best scores, {threshold: scores}, point labels)."""
        t = np.ascontiguousarray(truth_point_labels, np.uint32)
        if len(t) != self._n:
            raise ValueError("one ground-truth label per input point is required")
        cap = 4096
        ts = np.zeros(cap, np.float32); ps = (Performance * cap)()
        n = ctypes.c_size_t(); bt = ctypes.c_float(); bp = Performance()
        labels = np.empty(self._n, np.uint32)
        _check(self.lib, self.lib.f3ds_auto_threshold(self.handle, ctypes.byref(params), t.ctypes.data, start, end, step, ts.ctypes.data, ps, cap,
                                                      ctypes.byref(n), ctypes.byref(bt), ctypes.byref(bp), labels.ctypes.data, 0, ctypes.byref(self.result)))
        m = min(n.value, cap)
        return bt.value, bp, {float(ts[i]): ps[i].as_dict() for i in range(m)}, labels

    def voxel_centroid_cloud(self):
        """getVoxelCentroidCloud / getLabeledVoxelCloud: (xyz, rgba, supervoxel label) per voxel in leaf order."""
        n = ctypes.c_size_t()
        _check(self.lib, self.lib.f3ds_get_voxel_centroid_cloud(self.handle, None, None, None, 0, ctypes.byref(n)))
        xyz = np.zeros((n.value, 3), np.float32); rgba = np.zeros(n.value, np.uint32); lab = np.zeros(n.value, np.uint32)
        _check(self.lib, self.lib.f3ds_get_voxel_centroid_cloud(self.handle, xyz.ctypes.data, rgba.ctypes.data, lab.ctypes.data, n.value, ctypes.byref(n)))
        return xyz, rgba, lab

    def supervoxels(self):
        """The supervoxel_clusters map: dict of arrays label, xyz, rgb, normal, n_voxels (ascending label)."""
        n = ctypes.c_size_t()
        _check(self.lib, self.lib.f3ds_get_supervoxels(self.handle, None, None, None, None, None, 0, ctypes.byref(n)))
        k = n.value
        out = dict(label=np.zeros(k, np.uint32), xyz=np.zeros((k, 3), np.float32), rgb=np.zeros((k, 3), np.float32), normal=np.zeros((k, 3), np.float32),
                   n_voxels=np.zeros(k, np.uint32))
        _check(self.lib, self.lib.f3ds_get_supervoxels(self.handle, out["label"].ctypes.data, out["xyz"].ctypes.data, out["rgb"].ctypes.data, out["normal"].ctypes.data,
                                                       out["n_voxels"].ctypes.data, k, ctypes.byref(n)))
        return out

    def refine_supervoxels(self, num_itr):
        """refineSupervoxels(num_itr): dict with per-voxel ``voxel_label`` / ``voxel_normal`` (leaf order) and the refined
        supervoxel map ``label, xyz, rgb, normal, n_voxels``.  The frame's own supervoxels and clustering are untouched."""
        _check(self.lib, self.lib.f3ds_refine_supervoxels(self.handle, int(num_itr)))
        n = ctypes.c_size_t()
        _check(self.lib, self.lib.f3ds_get_refined_voxels(self.handle, None, None, 0, ctypes.byref(n)))
        vl = np.zeros(n.value, np.uint32); vn = np.zeros((n.value, 3), np.float32)
        _check(self.lib, self.lib.f3ds_get_refined_voxels(self.handle, vl.ctypes.data, vn.ctypes.data, n.value, ctypes.byref(n)))
        _check(self.lib, self.lib.f3ds_get_refined_supervoxels(self.handle, None, None, None, None, None, 0, ctypes.byref(n)))
        k = n.value
        out = dict(voxel_label=vl, voxel_normal=vn, label=np.zeros(k, np.uint32), xyz=np.zeros((k, 3), np.float32), rgb=np.zeros((k, 3), np.float32),
                   normal=np.zeros((k, 3), np.float32), n_voxels=np.zeros(k, np.uint32))
        _check(self.lib, self.lib.f3ds_get_refined_supervoxels(self.handle, out["label"].ctypes.data, out["xyz"].ctypes.data, out["rgb"].ctypes.data,
                                                               out["normal"].ctypes.data, out["n_voxels"].ctypes.data, k, ctypes.byref(n)))
        return out

    def supervoxel_adjacency(self):
        """getSupervoxelAdjacency after clear_adjacency: (E, 2) array of label pairs a < b, sorted."""
        n = ctypes.c_size_t()
        _check(self.lib, self.lib.f3ds_get_supervoxel_adjacency(self.handle, None, 0, ctypes.byref(n)))
        pairs = np.zeros((n.value, 2), np.uint32)
        _check(self.lib, self.lib.f3ds_get_supervoxel_adjacency(self.handle, pairs.ctypes.data, n.value, ctypes.byref(n)))
        return pairs

    def region_adjacency(self):
        """get_currentstate().second after cluster(): (K, 2) array of surviving supervoxel labels a < b, sorted."""
        n = ctypes.c_size_t()
        _check(self.lib, self.lib.f3ds_get_region_adjacency(self.handle, None, 0, ctypes.byref(n)))
        pairs = np.zeros((n.value, 2), np.uint32)
        _check(self.lib, self.lib.f3ds_get_region_adjacency(self.handle, pairs.ctypes.data, n.value, ctypes.byref(n)))
        return pairs

    def cluster_supervoxels(self, sv, adjacency_pairs, params):
        """f3ds_cluster_supervoxels: Clustering::set_initialstate(segm, adj) + cluster(threshold) on caller-supplied supervoxels.
        ``sv``: dict of arrays label, voxel_offset, voxel_xyz, voxel_rgba, centroid_xyz, normal (see pack_supervoxels);
        ``adjacency_pairs``: (P, 2) uint32 in multimap iteration order.  Returns (region_of_sv, voxel_labels)."""
        a = {k: np.ascontiguousarray(sv[k], np.float32 if k in ("voxel_xyz", "centroid_xyz", "normal") else np.uint32) for k in
             ("label", "voxel_offset", "voxel_xyz", "voxel_rgba", "centroid_xyz", "normal")}
        S = len(a["label"])
        # the C entry reads voxel_offset[0..S], voxel_offset[S] voxels and S centroid / normal rows through raw pointers: arrays that disagree with each other
        # would be read past their end (pack_supervoxels' output is consistent; any other dict is checked here)
        if a["voxel_offset"].shape != (S + 1,):
            raise ValueError("voxel_offset must hold n_supervoxels + 1 = %d entries, has shape %s" % (S + 1, a["voxel_offset"].shape))
        nvox = int(a["voxel_offset"][S])
        if a["voxel_xyz"].size != 3 * nvox or a["voxel_rgba"].size != nvox:
            raise ValueError("voxel_xyz / voxel_rgba must hold voxel_offset[-1] = %d voxels (have %d / %d)" % (nvox, a["voxel_xyz"].size // 3, a["voxel_rgba"].size))
        if a["centroid_xyz"].size != 3 * S or a["normal"].size != 3 * S:
            raise ValueError("centroid_xyz and normal must have shape (%d, 3)" % S)
        st = SupervoxelSet(S, *[a[k].ctypes.data for k in ("label", "voxel_offset", "voxel_xyz", "voxel_rgba", "centroid_xyz", "normal")])
        pairs = np.ascontiguousarray(adjacency_pairs, np.uint32).reshape(-1, 2)
        region = np.zeros(S, np.uint32); vlab = np.zeros(nvox, np.uint32)
        _check(self.lib, self.lib.f3ds_cluster_supervoxels(self.handle, ctypes.byref(st), pairs.ctypes.data, len(pairs), ctypes.byref(params), region.ctypes.data,
                                                           vlab.ctypes.data, ctypes.byref(self.result)))
        self._n = nvox
        return region, vlab

    def regions(self):
        """get_currentstate().first: dict of arrays label, n_voxels, xyz (centroid_), normal (normal_), rgb (mean_color) per merged region, ascending key."""
        n = ctypes.c_size_t()
        _check(self.lib, self.lib.f3ds_get_regions(self.handle, None, None, None, None, None, 0, ctypes.byref(n)))
        k = n.value
        out = dict(label=np.zeros(k, np.uint32), n_voxels=np.zeros(k, np.uint32), xyz=np.zeros((k, 3), np.float32), normal=np.zeros((k, 3), np.float32),
                   rgb=np.zeros((k, 3), np.float32))
        _check(self.lib, self.lib.f3ds_get_regions(self.handle, out["label"].ctypes.data, out["n_voxels"].ctypes.data, out["xyz"].ctypes.data, out["normal"].ctypes.data,
                                                   out["rgb"].ctypes.data, k, ctypes.byref(n)))
        return out

    def region_voxels(self):
        """The regions' voxels_ clouds concatenated in the order of regions(): (xyz, rgba, voxel index)."""
        n = ctypes.c_size_t()
        _check(self.lib, self.lib.f3ds_get_region_voxels(self.handle, None, None, None, 0, ctypes.byref(n)))
        xyz = np.zeros((n.value, 3), np.float32); rgba = np.zeros(n.value, np.uint32); idx = np.zeros(n.value, np.uint32)
        _check(self.lib, self.lib.f3ds_get_region_voxels(self.handle, xyz.ctypes.data, rgba.ctypes.data, idx.ctypes.data, n.value, ctypes.byref(n)))
        return xyz, rgba, idx

    def voxel_cloud(self):
        n = ctypes.c_size_t()
        _check(self.lib, self.lib.f3ds_get_voxel_cloud(self.handle, None, None, None, 0, ctypes.byref(n)))
        xyz = np.zeros((n.value, 3), np.float32); lab = np.zeros(n.value, np.uint32); rgba = np.zeros(n.value, np.uint32)
        _check(self.lib, self.lib.f3ds_get_voxel_cloud(self.handle, xyz.ctypes.data, lab.ctypes.data, rgba.ctypes.data, n.value, ctypes.byref(n)))
        return xyz, lab, rgba

    def merge_layout(self):
        """(waves per frame, per-edge arrays in LDS) of the merge kernel the last cluster stage ran; (0, 0) = d_merge (F3DS_DBG_MERGE_LAYOUT)."""
        nb = ctypes.c_size_t(); buf = np.zeros(2, np.uint32)
        _check(self.lib, self.lib.f3ds_get_debug(self.handle, 20, buf.ctypes.data, 8, ctypes.byref(nb)))
        return int(buf[0]), int(buf[1])

    def stage0_path(self):
        """'tiles' or 'sort': how the last frame was voxelised (F3DS_DBG_STAGE0_PATH)."""
        nb = ctypes.c_size_t(); buf = np.zeros(1, np.uint32)
        _check(self.lib, self.lib.f3ds_get_debug(self.handle, 23, buf.ctypes.data, 4, ctypes.byref(nb)))
        return "tiles" if buf[0] else "sort"

    def sweep_stats(self):
        """(full, incremental, fallback, skipped) sweeps of the last run of label-propagation sweeps (F3DS_DBG_SWEEP_STATS)."""
        nb = ctypes.c_size_t(); buf = np.zeros(4, np.uint32)
        _check(self.lib, self.lib.f3ds_get_debug(self.handle, 22, buf.ctypes.data, 16, ctypes.byref(nb)))
        return tuple(int(x) for x in buf)

    def tile_list_lengths(self):
        """Length of every 128-voxel tile's one-ring list; 0xFFFFFFFF = the tile overflowed the LDS tables (F3DS_DBG_TILE_LIST_LEN)."""
        nb = ctypes.c_size_t()
        _check(self.lib, self.lib.f3ds_get_debug(self.handle, 21, None, 0, ctypes.byref(nb)))
        buf = np.zeros(nb.value // 4, np.uint32)
        _check(self.lib, self.lib.f3ds_get_debug(self.handle, 21, buf.ctypes.data, nb.value, ctypes.byref(nb)))
        return buf

    def debug(self, name):
        nb = ctypes.c_size_t()
        _check(self.lib, self.lib.f3ds_get_debug(self.handle, DBG[name], None, 0, ctypes.byref(nb)))
        buf = np.zeros(nb.value, np.uint8)
        _check(self.lib, self.lib.f3ds_get_debug(self.handle, DBG[name], buf.ctypes.data, nb.value, ctypes.byref(nb)))
        return buf.view(DBG_DTYPE[name])


class FrameStream:
    """Frame pipeline (f3ds_stream_*): frames in from host memory, per-point labels out in submission order, up to
    ``depth`` frames in flight on ``groups`` host threads.  ``submit`` returns False instead of blocking when the
    pipeline is full; ``next`` returns (tag, labels, Result) of the oldest frame, None when nothing is in flight."""

    def __init__(self, device=0, depth=8, groups=0):
        self.lib = load_library()
        h = ctypes.c_void_p()
        _check(self.lib, self.lib.f3ds_stream_create(device, depth, groups, ctypes.byref(h)))
        self.handle = h
        self.depth = depth

    def close(self):
        if getattr(self, "handle", None):
            self.lib.f3ds_stream_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def pending(self):
        return self.lib.f3ds_stream_pending(self.handle)

    def buffer(self, n):
        """(n,4) float32 view of the pinned input buffer the next submit uses, or None when the pipeline is full."""
        p = ctypes.c_void_p()
        rc = self.lib.f3ds_stream_buffer(self.handle, n, ctypes.byref(p))
        if rc == ERR_BUSY:
            return None
        _check(self.lib, rc)
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_float)), shape=(max(n, 1), 4))[:n]

    def submit(self, points, params, tag=0):
        pts = np.ascontiguousarray(points, np.float32).reshape(-1, 4)
        rc = self.lib.f3ds_stream_submit(self.handle, pts.ctypes.data, len(pts), ctypes.byref(params), tag)
        if rc == ERR_BUSY:
            return False
        _check(self.lib, rc)
        return True

    def next(self, wait=True):
        n = ctypes.c_size_t(); tag = ctypes.c_uint64(); res = Result()
        probe = np.empty(1, np.uint32)
        rc = self.lib.f3ds_stream_next(self.handle, probe.ctypes.data, 0, ctypes.byref(n), ctypes.byref(tag), ctypes.byref(res), 1 if wait else 0)
        if rc in (ERR_EMPTY, ERR_BUSY):
            return None
        labels = np.empty(n.value, np.uint32)
        if rc == ERR_CAPACITY:                              # the frame is done and still there: now with room for its labels
            rc = self.lib.f3ds_stream_next(self.handle, labels.ctypes.data, len(labels), ctypes.byref(n), ctypes.byref(tag), ctypes.byref(res), 1)
        _check(self.lib, rc)
        return tag.value, labels, res

    def peek(self, wait=True):
        """(tag, labels, Result) of the oldest frame without copying: ``labels`` is a view of the slot's pinned buffer,
        valid until ``drop()``.  None when nothing is in flight (or, with wait=False, not done yet)."""
        n = ctypes.c_size_t(); tag = ctypes.c_uint64(); res = Result(); p = ctypes.c_void_p()
        rc = self.lib.f3ds_stream_peek(self.handle, ctypes.byref(p), ctypes.byref(n), ctypes.byref(tag), ctypes.byref(res), 1 if wait else 0)
        if rc in (ERR_EMPTY, ERR_BUSY):
            return None
        _check(self.lib, rc)
        lab = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint32)), shape=(max(n.value, 1),))[:n.value]
        return tag.value, lab, res

    def drop(self):
        """Take the oldest (finished) frame out of the pipeline without copying its labels."""
        rc = self.lib.f3ds_stream_next(self.handle, None, 0, None, None, None, 1)
        if rc != ERR_EMPTY:
            _check(self.lib, rc)

    def run(self, frames, params):
        """Generator: feed an iterable of (N,4) frames, yield (index, labels, Result) in order."""
        i = 0
        for pts in frames:
            while not self.submit(pts, params, i):
                yield self.next()
            i += 1
        while self.pending():
            yield self.next()


class MultiGpu:
    """f3ds_multi_*: a batch of independent frames sharded over the GPUs of one node from ONE process (frame i on
    devices[i mod G], one host thread per GPU inside the library), per-point labels gathered on devices[0] over RCCL and
    returned as host arrays.  The torch.distributed form of the same partitioning (one process per GPU) is batch.py."""

    def __init__(self, devices=None, n_devices=None, max_frames_per_device=8):
        self.lib = load_library()
        if devices is not None:
            arr = (ctypes.c_int * len(devices))(*devices); n = len(devices)
        else:
            arr = None; n = int(n_devices or 1)
        h = ctypes.c_void_p()
        rc = self.lib.f3ds_multi_create(arr, n, int(max_frames_per_device), ctypes.byref(h))
        if rc:
            raise F3dsError(rc, self.lib.f3ds_strerror(rc).decode() + " " + self.lib.f3ds_multi_last_error().decode())
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.lib.f3ds_multi_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def devices(self):
        return self.lib.f3ds_multi_devices(self.handle)

    def device_of_frame(self, frame):
        return self.lib.f3ds_multi_device_of_frame(self.handle, frame)

    def _error(self, rc):
        return F3dsError(rc, self.lib.f3ds_strerror(rc).decode() + " " + self.lib.f3ds_multi_last_error().decode() + " " + self.lib.f3ds_last_hip_error().decode())

    def reserve(self, max_points_per_frame):
        """Allocate the label blocks for full batches of frames of up to that many points now, not inside the first batches."""
        rc = self.lib.f3ds_multi_reserve(self.handle, int(max_points_per_frame))
        if rc:
            raise self._error(rc)

    def submit(self, frames, params):
        """Queue a batch (list of (N_i, 4) float32 arrays) and return a ticket: the GPUs compute it while the batch before it is
        gathered and copied out.  At most two batches are in flight (F3dsError ERR_BUSY: collect the older one first)."""
        k = len(frames)
        vp = ctypes.c_void_p
        arrs = [np.ascontiguousarray(f, np.float32).reshape(-1, 4) for f in frames]
        out = [np.empty(len(a), np.uint32) for a in arrs]
        pp = (vp * max(k, 1))(*[vp(a.ctypes.data) for a in arrs]); lp = (vp * max(k, 1))(*[vp(o.ctypes.data) for o in out])
        cnt = (ctypes.c_size_t * max(k, 1))(*[len(a) for a in arrs])
        results = (Result * max(k, 1))()
        prm = params.copy()
        t = ctypes.c_int(-1)
        rc = self.lib.f3ds_multi_submit(self.handle, pp, cnt, k, ctypes.byref(prm), lp, results, ctypes.byref(t))
        if rc:
            raise self._error(rc)
        if not hasattr(self, "_inflight"):
            self._inflight = {}
        self._inflight[t.value] = (arrs, out, results, k)      # the buffers stay alive until the batch is collected
        return t.value

    def collect(self, ticket):
        """Wait for a submitted batch.  Returns (list of label arrays, list of Result)."""
        arrs, out, results, k = self._inflight.pop(ticket)
        rc = self.lib.f3ds_multi_collect(self.handle, int(ticket))
        if rc:
            raise self._error(rc)
        return out, [results[i] for i in range(k)]

    def segment(self, frames, params):
        """frames: list of (N_i, 4) float32 arrays.  Returns (list of label arrays, list of Result)."""
        if not frames:
            return [], []
        return self.collect(self.submit(frames, params))


def segment_batch(ctxs, points, params, labels_out=None, n=None, on_device=False, raw_host=False):
    """Segment len(ctxs) independent frames at once (one context per frame, all on one GPU).
    Host mode: points = list of (N_i,4) float32 arrays, returns a list of label arrays.
    Device mode (on_device): points / labels_out = lists of device pointers, n = list of point counts.
    raw_host: the same with host pointers (e.g. pinned buffers of the caller): PCIe both ways inside the call."""
    lib = load_library()
    k = len(ctxs)
    vp = ctypes.c_void_p
    handles = (vp * k)(*[c.handle for c in ctxs])
    results = (Result * k)()
    if on_device or raw_host:
        where = 1 if on_device else 0
        pp = (vp * k)(*[vp(int(p)) for p in points]); lp = (vp * k)(*[vp(int(p)) for p in labels_out])
        cnt = (ctypes.c_size_t * k)(*[int(x) for x in n])
        _check(lib, lib.f3ds_segment_batch(handles, k, pp, cnt, where, ctypes.byref(params), lp, where, results))
        for c, x in zip(ctxs, n):
            c._n = int(x)
        out = None
    else:
        arrs = [np.ascontiguousarray(p, np.float32).reshape(-1, 4) for p in points]
        out = [np.empty(len(a), np.uint32) for a in arrs]
        pp = (vp * k)(*[vp(a.ctypes.data) for a in arrs]); lp = (vp * k)(*[vp(o.ctypes.data) for o in out])
        cnt = (ctypes.c_size_t * k)(*[len(a) for a in arrs])
        _check(lib, lib.f3ds_segment_batch(handles, k, pp, cnt, 0, ctypes.byref(params), lp, 0, results))
        for c, a in zip(ctxs, arrs):
            c._n = len(a)
    for c, r in zip(ctxs, results):
        ctypes.memmove(ctypes.byref(c.result), ctypes.byref(r), ctypes.sizeof(Result))
    return out


def segment(points, params=None, device=0):
    """XYZRGBA frame -> (per-point region ids, Result).  One-shot convenience wrapper."""
    ctx = Context(device)
    try:
        labels = ctx.segment(points, params or launch_params())
        res = Result()
        ctypes.memmove(ctypes.byref(res), ctypes.byref(ctx.result), ctypes.sizeof(Result))
        return labels, res
    finally:
        ctx.close()


# ---- mirror of the reference's classes -------------------------------------------------------------
class SupervoxelClustering:
    """The nine-call VCCS sequence of main() (supervoxel_clustering.cpp:348-367) as one object."""

    def __init__(self, voxel_resolution, seed_resolution, context=None):
        self.ctx = context or Context()
        self.params = default_params(voxel_res=voxel_resolution, seed_res=seed_resolution)
        self.cloud = None

    def setUseSingleCameraTransform(self, val):
        self.params.use_transform = 1 if val else 0

    def setInputCloud(self, points):
        self.cloud = np.ascontiguousarray(points, np.float32).reshape(-1, 4)

    def setColorImportance(self, v):
        self.params.w_color = v

    def setSpatialImportance(self, v):
        self.params.w_spatial = v

    def setNormalImportance(self, v):
        self.params.w_normal = v

    def extract(self):
        """extract(supervoxel_clusters) (:356): runs the frame and returns the supervoxels (see Context.supervoxels)."""
        if self.cloud is None:
            raise LogicError(-5, "setInputCloud first")
        self.ctx.segment(self.cloud, self.params)
        self._extracted = True
        return self.ctx.supervoxels()

    def getVoxelCentroidCloud(self):               # :359 -> (xyz, rgba)
        xyz, rgba, _ = self.ctx.voxel_centroid_cloud()
        return xyz, rgba

    def getLabeledVoxelCloud(self):                # voxel centroids with their supervoxel label
        xyz, _, lab = self.ctx.voxel_centroid_cloud()
        return xyz, lab

    def makeSupervoxelNormalCloud(self):           # :360 -> (centroid xyz, normal) per supervoxel
        sv = self.ctx.supervoxels()
        return sv["xyz"], sv["normal"]

    def refineSupervoxels(self, num_itr):          # :371 -> refined per-voxel labels / normals and supervoxel map
        return self.ctx.refine_supervoxels(num_itr)

    def getSupervoxelAdjacency(self):              # :365
        return self.ctx.supervoxel_adjacency()


class Clustering:
    """Mirror of ``class Clustering`` (clustering.h:84-212) on top of the device path."""

    def __init__(self, c=LAB_CIEDE00, g=NORMALS_DIFF, m=ADAPTIVE_LAMBDA):
        self.delta_c_type, self.delta_g_type = c, g
        self.set_merging(m)
        self._super = None
        self._labels = None

    def set_delta_c(self, d):
        self.delta_c_type = d

    def set_delta_g(self, d):
        self.delta_g_type = d

    def set_merging(self, m):                      # clustering.cpp:562-567
        self.merging_type, self.lambda_, self.bins_num = m, 0.5, 500

    def set_lambda(self, l):                       # clustering.cpp:574-582
        if self.merging_type != MANUAL_LAMBDA:
            raise LogicError(-5, "Lambda can be set only if the merging criterion is set to MANUAL_LAMBDA")
        if l < 0 or l > 1:
            raise ValueError("Argument outside range [0, 1]")
        self.lambda_ = l

    def set_bins_num(self, b):                     # clustering.cpp:589-597
        if self.merging_type != EQUALIZATION:
            raise LogicError(-5, "Bins number can be set only if the merging criterion is set to EQUALIZATION")
        if b < 0:
            raise ValueError("Argument lower than 0")
        self.bins_num = b

    def get_delta_c(self):
        return self.delta_c_type

    def get_delta_g(self):
        return self.delta_g_type

    def get_merging(self):
        return self.merging_type

    def get_lambda(self):
        return self.lambda_

    def get_bins_num(self):
        return self.bins_num

    def set_initialstate(self, segm, adj=None, context=None):
        """set_initialstate(segm, adj) (clustering.cpp:605-612).  Two forms:
        * ``segm`` = {label: dict(voxels_xyz, voxels_rgba, centroid, normal)}, ``adj`` = iterable of (first, second) in multimap
          iteration order -- supervoxels of ANY algorithm, as the reference's signature takes them (f3ds_cluster_supervoxels);
        * ``segm`` = a SupervoxelClustering object, ``adj`` omitted -- its extract() result is already device-resident."""
        if adj is None and isinstance(segm, SupervoxelClustering):
            self._super, self._user = segm, None
            self._segmented = False
            return
        if adj is None:
            raise TypeError("set_initialstate(segm, adj): the adjacency map is missing")
        self._user = (pack_supervoxels(segm), np.array(list(adj), np.uint32).reshape(-1, 2), context or Context())
        self._super = _UserState(self._user[2])
        self._segmented = False

    def _params(self, threshold):
        p = self._super.params.copy()
        p.color_metric, p.geom_metric, p.merging = self.delta_c_type, self.delta_g_type, self.merging_type
        p.lambda_ = self.lambda_ if self.merging_type == MANUAL_LAMBDA else 0.0
        p.bins = self.bins_num if self.merging_type == EQUALIZATION else 0
        p.threshold = threshold
        return p

    def cluster(self, threshold):                  # clustering.cpp:670-679
        if self._super is None:
            raise LogicError(-5, "Cannot call 'cluster' before setting an initial state with 'set_initialstate'")
        ctx = self._super.ctx
        if not self._segmented and getattr(self, "_user", None):
            self._region_of_sv, self._labels = ctx.cluster_supervoxels(self._user[0], self._user[1], self._params(threshold))
            self._segmented = True
        elif not self._segmented:
            self._labels = ctx.segment(self._super.cloud, self._params(threshold))
            self._segmented = True
        else:
            self._labels = ctx.recluster(self._params(threshold))
            if getattr(self, "_user", None):
                # a later cluster(t) moves the supervoxels to other regions: F3DS_DBG_SV_REGION lists the surviving label per supervoxel in ascending
                # label order, the caller's rows may be in any order
                lab = np.asarray(self._user[0]["label"], np.uint32)
                reg = ctx.debug("SV_REGION")
                out = np.zeros(len(lab), np.uint32)
                out[np.argsort(lab, kind="stable")] = reg
                self._region_of_sv = out
        if self.merging_type == ADAPTIVE_LAMBDA:
            self.lambda_ = ctx.result.lambda_

    def all_thresh(self, truth_point_labels, start_thresh, end_thresh, step_thresh):      # clustering.cpp:691-741
        """{threshold: performanceSet dict} over the sweep.  As in the reference the state is left clustered at the LAST threshold of the
        sweep (:718-726); IndexError (std::out_of_range, :694-698) for bounds outside [0, 1]; start > end are swapped (:699-705)."""
        if self._super is None:
            raise LogicError(-5, "Cannot call 'all_thresh' before setting an initial state with 'set_initialstate'")
        if getattr(self, "_user", None):
            raise LogicError(-5, "all_thresh needs the frame's points (ground truth is per input point): use a SupervoxelClustering state")
        if start_thresh < 0 or start_thresh > 1 or end_thresh < 0 or end_thresh > 1 or step_thresh < 0 or step_thresh > 1:
            raise IndexError("start_thresh, end_thresh and/or step_thresh outside of range [0, 1]")
        ctx = self._super.ctx
        if not self._segmented:
            ctx.segment(self._super.cloud, self._params(min(start_thresh, end_thresh)))
            self._segmented = True
        bt, bp, table, self._labels = ctx.auto_threshold(self._params(0.0), truth_point_labels, start_thresh, end_thresh, step_thresh)
        if table:
            self._labels = ctx.recluster(self._params(max(table)))      # (f3ds_auto_threshold leaves the context at the best threshold: main()'s use)
        if self.merging_type == ADAPTIVE_LAMBDA:
            self.lambda_ = ctx.result.lambda_
        return table

    @staticmethod
    def best_thresh(all_performances):             # clustering.cpp:748-774: first strictly greater F-score, (0, zeros) if none
        best_t, best_p = 0.0, {k: 0.0 for k in ("voi", "precision", "recall", "fscore", "wov", "fpr", "fnr")}
        for t in sorted(all_performances):
            if all_performances[t]["fscore"] > best_p["fscore"]:
                best_t, best_p = t, all_performances[t]
        return best_t, best_p

    def eval_performance(self, truth_point_labels):
        """Testing(get_labeled_cloud(), truth_cloud).eval_performance() (src/supervoxel_clustering.cpp:462-463)."""
        return self._super.ctx.evaluate(truth_point_labels).as_dict()

    def get_labeled_cloud(self):                   # clustering.cpp:640-663 -> (xyz, label)
        xyz, lab, _ = self._super.ctx.voxel_cloud()
        return xyz, lab

    def get_colored_cloud(self):                   # clustering.cpp:631-633 -> (xyz, rgba)
        xyz, _, rgba = self._super.ctx.voxel_cloud()
        return xyz, rgba

    @staticmethod
    def label2color(xyz, labels):                  # clustering.cpp:793-812 -> (xyz, rgba): the lookup-table colour of every label, opaque alpha
        labels = np.asarray(labels, np.uint32)
        table = np.array([label_color(i) for i in range(256)], np.uint32)
        return np.asarray(xyz, np.float32), (table[labels % 256] | np.uint32(0xFF000000)).astype(np.uint32)

    @staticmethod
    def color2label(xyz, rgba):                    # clustering.cpp:824-846 -> (xyz, label): one label per distinct colour, numbered in order of first appearance
        rgba = np.asarray(rgba, np.uint32)
        uniq, first, inv = np.unique(rgba, return_index=True, return_inverse=True)
        rank = np.empty(len(uniq), np.uint32)
        rank[np.argsort(first, kind="stable")] = np.arange(len(uniq), dtype=np.uint32)
        return np.asarray(xyz, np.float32), rank[inv].astype(np.uint32)

    def get_region_of_supervoxel(self):
        """After set_initialstate(segm, adj) + cluster(t): label of the region every input supervoxel ended in at the LAST threshold, in the row
        order of the supervoxel arrays."""
        return getattr(self, "_region_of_sv", None)

    def get_point_labels(self):
        """Per input point region id (the composition with pcl getLabeledCloud); per input VOXEL after
        set_initialstate(segm, adj)."""
        return self._labels

    def get_currentstate(self):                    # clustering.cpp:619-624
        """(segments, adjacency): segments = {label: dict(voxels_xyz, voxels_rgba, voxel_index, centroid, normal, mean_rgb)} of the merged
        regions (state.segments), adjacency = (K, 2) array of label pairs a < b (weight2adj(state.weight_map))."""
        ctx = self._super.ctx
        r = ctx.regions()
        xyz, rgba, idx = ctx.region_voxels()
        segm, o = {}, 0
        for k in range(len(r["label"])):
            n = int(r["n_voxels"][k])
            segm[int(r["label"][k])] = dict(voxels_xyz=xyz[o:o + n], voxels_rgba=rgba[o:o + n], voxel_index=idx[o:o + n], centroid=r["xyz"][k],
                                            normal=r["normal"][k], mean_rgb=r["rgb"][k])
            o += n
        return segm, ctx.region_adjacency()


class _UserState:
    """What Clustering keeps in place of a SupervoxelClustering object after set_initialstate(segm, adj)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.params = default_params()
        self.cloud = None
