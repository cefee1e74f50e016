#!/usr/bin/env python3
"""Compare a dump of the REAL reference (tests/pcl_pin/pcl_dump.cpp, built by a maintainer who has PCL >= 1.8 + OpenCV) with this repo's CPU oracle
-- and, when a GPU is visible, with the HIP path -- array by array.  One command closes the "parity unpinned" half of SURVEY.md 8c:

    python tests/pcl_pin/compare_with_pcl.py <dump dir> <frame.pcd> [-v 0.008 -s 0.08 ... the flags pcl_dump ran with] [--gpu]

Two independent pins come out of one dump:
  (1) the VCCS half: GRID, VOXEL_COUNT / XYZ / RGB / NORMAL / DIST, neighbour sets, supervoxel label per voxel and per point,
      supervoxel centroids, adjacency -- oracle (restated from SURVEY Appendix A) against PCL itself;
  (2) the merge half on PCL's OWN supervoxels (a10-a22, side-stepping (1) entirely): the dumped supervoxel map + adjacency go through
      f3ds_oracle_cluster_supervoxels (and f3ds_cluster_supervoxels with --gpu) and must reproduce PCL-fed Clustering's labelled cloud, regions,
      region adjacency and lambda;
plus LAB17 (OpenCV's float RGB -> Lab against the analytic restatement, SURVEY a13: expected to differ by up to ~1e-1 Lab, reported as max |diff|)
and GLASBEY (pcl::GlasbeyLUT against csrc/f3ds_glasbey.h: paste the dumped table there to make get_colored_cloud's colours the reference's).

`--self-test <dir>` writes a dump in the same format FROM THE ORACLE for a synthetic frame and compares it with itself: what the CPU test suite
runs, since no PCL exists in this image (tests/test_pcl_pin.py).  Test infrastructure: nothing in the product imports this."""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
DTYPES = {"f64": np.float64, "f32": np.float32, "u32": np.uint32, "i32": np.int32}


def read_dump(d):
    out = {}
    for line in open(os.path.join(d, "manifest.txt")):
        name, dt, cnt = line.split()
        out[name] = np.fromfile(os.path.join(d, name + ".bin"), DTYPES[dt], int(cnt))
    return out


def write_dump(d, arrays):
    os.makedirs(d, exist_ok=True)
    inv = {np.dtype(v): k for k, v in DTYPES.items()}
    with open(os.path.join(d, "manifest.txt"), "w") as m:
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            a.tofile(os.path.join(d, name + ".bin"))
            m.write("%s %s %d\n" % (name, inv[a.dtype], a.size))


def params_from_flags(P, flags):
    kw, i = dict(merging=P.ADAPTIVE_LAMBDA, threshold=0.2), 0
    names = {"-v": "voxel_res", "-s": "seed_res", "-c": "w_color", "-z": "w_spatial", "-n": "w_normal", "-t": "threshold"}
    while i < len(flags):
        a = flags[i]
        if a in names:
            kw[names[a]] = float(flags[i + 1]); i += 1
        elif a == "--NT":
            kw["use_transform"] = 0
        elif a == "--RGB":
            kw["color_metric"] = P.RGB_EUCL
        elif a == "--CVX":
            kw["geom_metric"] = P.CONVEX_NORMALS_DIFF
        elif a == "--ML":
            kw["merging"] = P.MANUAL_LAMBDA; kw["lambda"] = float(flags[i + 1]); i += 1
        elif a == "--EQ":
            kw["merging"] = P.EQUALIZATION; kw["bins"] = int(flags[i + 1]); i += 1
        elif a == "--PCL18":
            kw["leaf_order"] = 1
        i += 1
    return P.default_params(**kw)


def oracle_arrays(oracle, P, pts, prm):
    """The dump pcl_dump.cpp would write, from the oracle (names and layouts of its header comment)."""
    rc, labels, res, h = oracle.segment(pts, prm)
    assert rc == 0
    sv, pairs = h.export_supervoxels()
    nbr = h.get("VOXEL_NEIGHBORS").reshape(-1, 27)
    cloud = h.voxel_cloud(); reg = h.regions()
    svl = h.get("VOXEL_SVLABEL"); pv = h.get("POINT_VOXEL")
    lab17 = np.zeros((17 ** 3, 3), np.float32)
    lv = [0, 16, 32, 48, 64, 80, 96, 112, 128, 144, 160, 176, 192, 208, 224, 240, 255]
    k = 0
    fn = oracle.fn("rgb2lab")
    for r in lv:
        for g in lv:
            for b in lv:
                c = np.array([r, g, b], np.float32); o = np.zeros(3, np.float32)
                fn(ctypes.c_void_p(c.ctypes.data), ctypes.c_void_p(o.ctypes.data)); lab17[k] = o; k += 1
    # region adjacency from the initial edges and the surviving labels (weight2adj(state.weight_map))
    lut = dict(zip(h.get("SV_LABELS").tolist(), h.get("SV_REGION").tolist()))
    ra = sorted({(min(lut[int(a)], lut[int(b)]), max(lut[int(a)], lut[int(b)])) for a, b in h.get("EDGES").reshape(-1, 2) if lut[int(a)] != lut[int(b)]})
    return dict(GRID=h.get("GRID"), VOXEL_COUNT=h.get("VOXEL_COUNT"), VOXEL_XYZ=h.get("VOXEL_XYZ"), VOXEL_RGB=h.get("VOXEL_RGB"), VOXEL_NORMAL=h.get("VOXEL_NORMAL"),
                VOXEL_DIST=h.get("VOXEL_DIST"), VOXEL_NEIGHBOR_LIST=np.ascontiguousarray(nbr, np.int32), VOXEL_SVLABEL=svl,
                POINT_SVLABEL=np.where(pv >= 0, svl[np.maximum(pv, 0)], 0).astype(np.uint32), SV_LABELS=h.get("SV_LABELS"), SV_CENTROID=h.get("SV_CENTROID"),
                SV_VOXEL_OFFSET=sv["voxel_offset"], SV_VOXEL_XYZ=sv["voxel_xyz"], SV_VOXEL_RGBA=sv["voxel_rgba"], ADJACENCY=pairs,
                LAMBDA=np.array([res.lambda_], np.float32), CLOUD_XYZ=cloud[0], CLOUD_LABEL=cloud[1], REGION_LABELS=reg["label"], REGION_COUNT=reg["n_voxels"],
                REGION_CENTROID=reg["xyz"], REGION_NORMAL=reg["normal"], REGION_ADJACENCY=np.array(ra, np.uint32).reshape(-1, 2), LAB17=lab17,
                GLASBEY=np.array([P.label_color(i) for i in range(256)], np.uint32)), h


def report(name, want, got, rows):
    """One line per array: identical / differs (count, first index, max |diff| for floats).  Returns True when identical."""
    want, got = np.asarray(want).reshape(-1), np.asarray(got).reshape(-1)
    if want.shape != got.shape:
        rows.append((name, "SHAPE", "%d vs %d elements" % (want.size, got.size))); return False
    if want.dtype.kind == "f":
        nan = np.isnan(want) & np.isnan(got)
        same = (want.view(np.uint32 if want.dtype == np.float32 else np.uint64) == got.view(np.uint32 if got.dtype == np.float32 else np.uint64)) | nan
    else:
        same = want == got
    if same.all():
        rows.append((name, "identical", "%d elements" % want.size)); return True
    bad = np.nonzero(~same)[0]
    extra = ""
    if want.dtype.kind == "f":
        with np.errstate(invalid="ignore"):
            extra = ", max |diff| %.3g" % float(np.nanmax(np.abs(want.astype(np.float64) - got.astype(np.float64))))
    rows.append((name, "DIFFERS", "%d of %d elements, first at %d: %r vs %r%s" % (len(bad), want.size, bad[0], want[bad[0]], got[bad[0]], extra)))
    return False


def compare(dump, P, oracle, pts, prm, use_gpu=False):
    rows, ok = [], True
    mine, h = oracle_arrays(oracle, P, pts, prm)
    # (1) the VCCS half against PCL
    for name in ("GRID", "VOXEL_COUNT", "VOXEL_XYZ", "VOXEL_RGB", "VOXEL_NORMAL", "VOXEL_DIST", "VOXEL_SVLABEL", "POINT_SVLABEL", "SV_LABELS", "SV_CENTROID", "SV_VOXEL_OFFSET",
                 "SV_VOXEL_XYZ", "SV_VOXEL_RGBA"):
        if name in dump:
            ok &= report("vccs." + name, dump[name], mine[name], rows)
    if "VOXEL_NEIGHBOR_LIST" in dump and dump["VOXEL_NEIGHBOR_LIST"].size == mine["VOXEL_NEIGHBOR_LIST"].size:
        a = np.sort(dump["VOXEL_NEIGHBOR_LIST"].reshape(-1, 27), axis=1); b = np.sort(mine["VOXEL_NEIGHBOR_LIST"].reshape(-1, 27), axis=1)
        ok &= report("vccs.VOXEL_NEIGHBORS (as sets)", a, b, rows)
    if "ADJACENCY" in dump:
        a = dump["ADJACENCY"].reshape(-1, 2); b = mine["ADJACENCY"].reshape(-1, 2)
        ok &= report("vccs.ADJACENCY (sorted)", a[np.lexsort((a[:, 1], a[:, 0]))], b[np.lexsort((b[:, 1], b[:, 0]))], rows)
    # (2) the merge half on the DUMPED supervoxels (PCL's own when the dump is pcl_dump's)
    if all(k in dump for k in ("SV_LABELS", "SV_VOXEL_OFFSET", "SV_VOXEL_XYZ", "SV_VOXEL_RGBA", "SV_CENTROID", "ADJACENCY")):
        c = dump["SV_CENTROID"].reshape(-1, 10)
        sv = dict(label=dump["SV_LABELS"], voxel_offset=dump["SV_VOXEL_OFFSET"], voxel_xyz=dump["SV_VOXEL_XYZ"].reshape(-1, 3), voxel_rgba=dump["SV_VOXEL_RGBA"],
                  centroid_xyz=np.ascontiguousarray(c[:, 0:3]), normal=np.ascontiguousarray(c[:, 6:9]))
        pairs = dump["ADJACENCY"].reshape(-1, 2)
        engines = [("oracle", None)]
        if use_gpu:
            engines.append(("gpu", P.Context(0)))
        for tag, ctx in engines:
            if ctx is None:
                rc, region, vlab, res, h2 = oracle.cluster_supervoxels(sv, pairs, prm)
                assert rc == 0, rc
                cloud = h2.voxel_cloud(); reg = h2.regions(); lam = res.lambda_
                lut = dict(zip(sv["label"].tolist(), region.tolist()))
            else:
                region, vlab = ctx.cluster_supervoxels(sv, pairs, prm)
                cloud = ctx.voxel_cloud(); reg = ctx.regions(); lam = ctx.result.lambda_
                lut = dict(zip(sv["label"].tolist(), region.tolist()))
            kept = pairs[pairs[:, 0] < pairs[:, 1]]
            ra = sorted({(min(lut[int(a)], lut[int(b)]), max(lut[int(a)], lut[int(b)])) for a, b in kept if lut[int(a)] != lut[int(b)]})
            for name, got in (("LAMBDA", np.array([lam], np.float32)), ("CLOUD_XYZ", cloud[0]), ("CLOUD_LABEL", cloud[1]), ("REGION_LABELS", reg["label"]),
                              ("REGION_COUNT", reg["n_voxels"]), ("REGION_CENTROID", reg["xyz"]), ("REGION_NORMAL", reg["normal"]),
                              ("REGION_ADJACENCY", np.array(ra, np.uint32).reshape(-1, 2))):
                if name in dump:
                    ok &= report("merge[%s].%s" % (tag, name), dump[name], got, rows)
    # (3) colour tables
    if "LAB17" in dump:
        report("color.LAB17 (OpenCV float Lab vs the analytic formula: a difference up to ~1e-1 is expected, SURVEY a13)", dump["LAB17"], mine["LAB17"], rows)
    if "GLASBEY" in dump:
        report("color.GLASBEY (pcl::GlasbeyLUT vs csrc/f3ds_glasbey.h: cosmetic)", dump["GLASBEY"], mine["GLASBEY"], rows)
    return ok, rows


def main(argv):
    from conftest import CpuChecker, pkg
    P = pkg()
    oracle = CpuChecker(os.path.join(ROOT, "oracle", "libf3ds_oracle.so"), "f3ds_oracle")
    if argv and argv[0] == "--self-test":
        d = argv[1]
        pts = P.synth_frame(0, 7, 160, 120, 30)
        prm = params_from_flags(P, ["-v", "0.02", "-s", "0.2", "--CVX"])
        write_dump(d, oracle_arrays(oracle, P, pts, prm)[0])
        ok, rows = compare(read_dump(d), P, oracle, pts, prm, use_gpu="--gpu" in argv)
    else:
        if len(argv) < 2:
            print(__doc__); return 2
        d, pcd, flags = argv[0], argv[1], argv[2:]
        pts = P.read_pcd(pcd)
        prm = params_from_flags(P, flags)
        prm.fold_negative_z = 1
        ok, rows = compare(read_dump(d), P, oracle, pts, prm, use_gpu="--gpu" in flags)
    for name, verdict, detail in rows:
        print("%-9s %-40s %s" % (verdict, name, detail))
    print("RESULT:", "every pinned array identical" if ok else "differences found (see above)")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
