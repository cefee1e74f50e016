"""PCD reader (f3ds_pcd_read, the replacement for pcl::io::loadPCDFile at src/supervoxel_clustering.cpp:313) against
files written here in every layout PCL produces: any field order, extra fields of other types and counts, rgb as packed
float or rgba as uint32, label present or not, ascii / binary / binary_compressed (LZF, field-major); and malformed
files, which must give an error, not a crash."""
import struct

import numpy as np
import pytest

TYPES = {("F", 4): "<f4", ("F", 8): "<f8", ("U", 1): "<u1", ("U", 2): "<u2", ("U", 4): "<u4", ("I", 1): "<i1", ("I", 2): "<i2", ("I", 4): "<i4"}


def lzf_literal(data):
    """A valid LZF stream made of literal runs only (control byte n-1 < 32, then n bytes)."""
    out = bytearray()
    for i in range(0, len(data), 32):
        chunk = data[i:i + 32]
        out.append(len(chunk) - 1); out += chunk
    return bytes(out)


def make_pcd(rng, n, mode):
    fields = [("x", "F", 4, 1), ("y", "F", 4, 1), ("z", "F", 4, 1)]
    color = rng.choice(["rgb", "rgba", None])
    if color == "rgb":
        fields.append(("rgb", "F", 4, 1))
    elif color == "rgba":
        fields.append(("rgba", "U", 4, 1))
    has_label = bool(rng.integers(0, 2))
    if has_label:
        fields.append(("label", "U", 4, 1))
    for k in range(int(rng.integers(0, 4))):               # fields the reader has to skip
        t, s = list(TYPES)[int(rng.integers(0, len(TYPES)))]
        fields.append(("extra%d" % k, t, s, int(rng.choice([1, 1, 3]))))
    order = rng.permutation(len(fields))
    fields = [fields[i] for i in order]
    cols = {}
    for name, t, s, c in fields:
        if name in ("x", "y", "z"):
            v = rng.uniform(-3, 3, (n, 1)).astype(np.float32)
            v[rng.random(n) < 0.05] = np.nan
        elif name in ("rgb", "rgba"):
            v = rng.integers(0, 1 << 24, (n, 1)).astype(np.uint32)
            if name == "rgb":
                v = v.view(np.float32)
        elif name == "label":
            v = rng.integers(0, 5000, (n, 1)).astype(np.uint32)
        elif t == "F":
            v = rng.uniform(-9, 9, (n, c)).astype(TYPES[(t, s)])
        else:
            info = np.iinfo(np.dtype(TYPES[(t, s)]))
            v = rng.integers(max(info.min, -1000), min(info.max, 1000), (n, c)).astype(TYPES[(t, s)])
        cols[name] = v
    head = "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS %s\nSIZE %s\nTYPE %s\nCOUNT %s\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n" % (
        " ".join(f[0] for f in fields), " ".join(str(f[2]) for f in fields), " ".join(f[1] for f in fields), " ".join(str(f[3]) for f in fields), n, n, mode)
    if mode == "ascii":
        lines = []
        for i in range(n):
            parts = []
            for name, t, s, c in fields:
                for j in range(c):
                    x = cols[name][i, j]
                    parts.append("nan" if (t == "F" and np.isnan(x)) else (repr(float(x)) if t == "F" and name != "rgb" else ("%.9g" % x if t == "F" else str(int(x)))))
            lines.append(" ".join(parts))
        body = ("\n".join(lines) + "\n").encode()
    elif mode == "binary":
        rec = np.zeros(n, dtype=[(f[0], TYPES[(f[1], f[2])], (f[3],)) for f in fields])
        for name, *_ in fields:
            rec[name] = cols[name]
        body = rec.tobytes()
    else:
        raw = b"".join(np.ascontiguousarray(cols[f[0]][:, j]).tobytes() for f in fields for j in range(f[3]))     # field-major, component by component
        comp = lzf_literal(raw)
        body = struct.pack("<II", len(comp), len(raw)) + comp
    xyz = np.concatenate([cols["x"], cols["y"], cols["z"]], axis=1)
    rgba = cols[color].view(np.uint32)[:, 0] if color else np.zeros(n, np.uint32)
    lab = cols["label"][:, 0] if has_label else np.zeros(n, np.uint32)
    return head.encode() + body, xyz, rgba, lab, (color == "rgb" and mode == "ascii")


@pytest.mark.parametrize("mode", ["ascii", "binary", "binary_compressed"])
def test_random_layouts(P, tmp_path, mode):
    rng = np.random.default_rng({"ascii": 1, "binary": 2, "binary_compressed": 3}[mode])
    for it in range(25):
        n = int(rng.choice([1, 2, 17, 300]))
        data, xyz, rgba, lab, lossy_rgb = make_pcd(rng, n, mode)
        f = tmp_path / ("f%d.pcd" % it)
        f.write_bytes(data)
        pts, got_lab = P.read_pcd(str(f), with_labels=True)
        assert pts.shape == (n, 4), it
        assert np.array_equal(np.isnan(pts[:, :3]), np.isnan(xyz)) and np.array_equal(np.nan_to_num(pts[:, :3]), np.nan_to_num(xyz)), it
        if not lossy_rgb:        # an rgb float printed in ascii goes through decimal text (PCL has the same weakness)
            assert np.array_equal(pts[:, 3].copy().view(np.uint32), rgba), it
        assert np.array_equal(got_lab, lab), it


def test_malformed_files_are_refused(P, tmp_path):
    rng = np.random.default_rng(9)
    good, *_ = make_pcd(rng, 50, "binary")
    comp, *_ = make_pcd(rng, 50, "binary_compressed")
    cases = {
        "empty": b"",
        "header_only": good[:good.index(b"DATA")],
        "truncated_body": good[:-100],
        "truncated_compressed": comp[:-40],
        "bad_sizes": good.replace(b"SIZE 4", b"SIZE 0", 1),
        "no_xyz": good.replace(b" x ", b" q ", 1) if b" x " in good else good.replace(b"FIELDS x", b"FIELDS q", 1),
        "huge_points": good.replace(b"POINTS 50", b"POINTS 4000000000").replace(b"WIDTH 50", b"WIDTH 4000000000"),
        # header ends at "DATA binary" / "DATA binary_compressed" with no newline and no payload (size underflow)
        "no_newline_after_data": good[:good.index(b"DATA binary") + 11].replace(b"POINTS 50", b"POINTS 100000").replace(b"WIDTH 50", b"WIDTH 100000"),
        "no_newline_after_data_compressed": comp[:comp.index(b"DATA binary_compressed") + 22],
        "rgb_two_bytes": b"VERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 2\nTYPE F F F U\nCOUNT 1 1 1 1\nWIDTH 2\nHEIGHT 1\nPOINTS 2\nDATA binary\n" + bytes(28),
        "garbage": bytes(rng.integers(0, 256, 4096).astype(np.uint8)),
        "lzf_bad_backref": comp[:comp.index(b"DATA binary_compressed\n") + 23] + struct.pack("<II", 8, 5000) + bytes([0xE0, 0xFF, 0xFF, 0, 0, 0, 0, 0]),
    }
    for name, data in cases.items():
        f = tmp_path / (name + ".pcd")
        f.write_bytes(data)
        with pytest.raises(Exception):
            P.read_pcd(str(f))
