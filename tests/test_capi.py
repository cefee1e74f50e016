"""The C-ABI library loads without a GPU, exports every symbol include/f3ds.h declares, and its
host-side pieces (PCD i/o, synthetic frames, parameter defaults, error strings, CLI) behave."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import FIXTURE_PCD, ROOT


def test_every_declared_symbol_is_exported(P):
    lib = P.load_library()
    hdr = open(os.path.join(ROOT, "include", "f3ds.h")).read()
    names = set(re.findall(r"\b(f3ds_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), n
    assert lib.f3ds_version() == 120
    assert lib.f3ds_version_string().decode().startswith("f3ds 1.2.0 src:")


def test_merge_kernel_lds_layout(P):
    """The LDS carve-up of d_merge_il_t (host arithmetic shared with the kernel): what fits a compute unit, and how many voxel rows the speculative second merge
    of an epoch gets -- 128 with four waves; 1024, 512, 256 or 128 with eight, whichever the layout has room for (BASELINE config 4's supervoxels of ~190 voxels need > 128)."""
    limit = 160 * 1024 - 2560
    assert P.merge_layout_info(9305, 8, 2)[1] == 512 and P.merge_layout_info(9305, 8, 2)[3]            # the 1M-point bench frame, arrays in LDS
    assert P.merge_layout_info(25023, 8, 0)[1] == 1024 and P.merge_layout_info(25023, 8, 0)[3]        # config 4, arrays in global memory
    assert not P.merge_layout_info(25023, 8, 2)[3]                                                      # ... which do not fit LDS
    last = {}
    for waves in (4, 8):
        for res in (0, 2):
            for e in list(range(0, 2000, 37)) + list(range(2000, 40000, 331)):
                lds, rows, ecap, fits = P.merge_layout_info(e, waves, res)
                assert rows in ((128,) if waves == 4 else (128, 256, 512, 1024)) and ecap >= max(e, 64) and ecap % 64 == 0
                assert fits == (lds <= limit)
                if waves == 8 and rows > 128:
                    assert fits                                                   # a bigger second merge never costs the layout its place
                    assert lds + rows * 52 > limit or rows == 1024      # ... and the next size up would not have fitted
                prev = last.get((waves, res))
                assert prev is None or rows <= prev                               # more adjacencies never buy more rows
                last[(waves, res)] = rows
    with pytest.raises(P.F3dsError):
        P.merge_layout_info(100, 6, 2)


def test_struct_layout_matches_header(P):
    assert ctypes.sizeof(P.Params) == 14 * 4
    assert ctypes.sizeof(P.Result) == 96          # 92 bytes of fields, 8-byte aligned
    p = P.default_params()
    assert (round(p.voxel_res, 6), round(p.seed_res, 6), round(p.w_color, 6), round(p.w_spatial, 6), p.w_normal) == (0.008, 0.08, 0.2, 0.4, 1.0)
    assert (p.use_transform, p.color_metric, p.geom_metric, p.merging, p.fold_negative_z, p.leaf_order) == (1, 0, 0, 1, 1, 0)


def test_error_strings(P):
    lib = P.load_library()
    for code in range(0, -13, -1):
        assert lib.f3ds_strerror(code).decode() != "unknown error"
    assert lib.f3ds_strerror(-99).decode() == "unknown error"
    assert "Cannot call 'cluster'" in lib.f3ds_strerror(-5).decode()


def test_no_cpu_fallback(P):
    if P.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(P.F3dsError) as e:
        P.Context(0)
    assert e.value.code == -2


def test_frame_pipeline_argument_checks_need_no_gpu(P):
    lib = P.load_library()
    h = ctypes.c_void_p()
    assert lib.f3ds_stream_create(0, 0, 0, ctypes.byref(h)) == P.ERR_ARG and not h.value
    assert lib.f3ds_stream_create(0, 4, -1, ctypes.byref(h)) == P.ERR_ARG
    assert lib.f3ds_stream_create(0, 4, 0, None) == P.ERR_ARG
    prm = P.default_params()
    assert lib.f3ds_stream_submit(None, None, 0, ctypes.byref(prm), 0) == P.ERR_ARG
    assert lib.f3ds_stream_next(None, None, 0, None, None, None, 0) == P.ERR_ARG and lib.f3ds_stream_pending(None) == P.ERR_ARG
    lib.f3ds_stream_destroy(None)
    if P.device_count() == 0:
        with pytest.raises(P.F3dsError) as e:
            P.FrameStream(0, depth=2)
        assert e.value.code == P.ERR_NO_DEVICE


def test_pcd_roundtrip_all_modes(P, tmp_path):
    pts = P.read_pcd(FIXTURE_PCD)                       # binary_compressed (LZF, field-major)
    assert pts.shape == (307200, 4) and np.isfinite(pts[:, 0]).sum() == 241407
    sub = pts[100000:100500].copy()
    xyz = sub[:, :3].copy(); rgba = sub[:, 3].copy().view(np.uint32); lab = np.arange(500, dtype=np.uint32)
    for binary in (True, False):
        f = str(tmp_path / ("b.pcd" if binary else "a.pcd"))
        P.write_pcd(f, xyz, rgba, lab, binary=binary)
        back, blab = P.read_pcd(f, with_labels=True)
        assert np.array_equal(back.view(np.uint32), sub.view(np.uint32)) and np.array_equal(blab, lab)
    with pytest.raises(P.F3dsError):
        P.read_pcd(str(tmp_path / "missing.pcd"))


def test_synthetic_frames_are_deterministic(P):
    a = P.synth_frame(0, 1000, 64, 48, 30); b = P.synth_frame(0, 1000, 64, 48, 30); c = P.synth_frame(0, 1001, 64, 48, 30)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and not np.array_equal(a.view(np.uint32), c.view(np.uint32))
    assert 0 < np.isnan(a[:, 2]).sum() < 0.1 * len(a) and np.nanmin(a[:, 2]) > 0.25 and np.nanmax(a[:, 2]) < 3.3
    f = P.synth_frame(1, 3000, 100, 100, 0)
    assert np.isfinite(f[:, :3]).all() and f[:, 2].min() > 0


def test_label_colors(P):
    cols = {P.label_color(i) for i in range(256)}
    assert len(cols) == 256 and P.label_color(5) == P.label_color(5 + 256)


def test_cli_argument_contract():
    exe = os.path.join(ROOT, "fast-3d-pointcloud-segmentation_amd", "supervoxel_clustering")
    assert os.path.exists(exe)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "Syntax is:" in r.stdout                       # argc < 3 -> usage, exit 1
    r = subprocess.run([exe, "-t", "0.2", "--AL"], capture_output=True, text=True)
    assert r.returncode == 1 and "No input file or directory specified" in r.stderr
    r = subprocess.run([exe, "-p", FIXTURE_PCD, "-t", "0.2", "--AL", "--ML"], capture_output=True, text=True)
    assert r.returncode == 1 and "Only one parameter between --ML --AL and --EQ" in r.stderr
    r = subprocess.run([exe, "-d", "/nonexistent_dir", "-t", "0.2"], capture_output=True, text=True)
    assert r.returncode == 1 and "Specified directory" in r.stderr


def test_clustering_mirror_error_behaviour(P):
    c = P.Clustering()
    assert (c.get_delta_c(), c.get_delta_g(), c.get_merging(), c.get_lambda(), c.get_bins_num()) == (P.LAB_CIEDE00, P.NORMALS_DIFF, P.ADAPTIVE_LAMBDA, 0.5, 500)
    with pytest.raises(P.LogicError):
        c.set_lambda(0.3)                     # only under MANUAL_LAMBDA (clustering.cpp:575-577)
    with pytest.raises(P.LogicError):
        c.set_bins_num(10)
    with pytest.raises(P.LogicError):
        c.cluster(0.5)                        # before set_initialstate (clustering.cpp:671-673)
    c.set_merging(P.MANUAL_LAMBDA)
    with pytest.raises(ValueError):
        c.set_lambda(1.5)
    c.set_lambda(0.25); assert c.get_lambda() == 0.25
    c.set_merging(P.EQUALIZATION); assert c.get_lambda() == 0.5 and c.get_bins_num() == 500
    with pytest.raises(ValueError):
        c.set_bins_num(-1)


def test_multi_gpu_driver_argument_checks_need_no_gpu(P):
    """f3ds_multi_*: argument errors and the no-device answer come back as codes, never as a crash, without any GPU work."""
    lib = P.load_library()
    h = ctypes.c_void_p()
    assert lib.f3ds_multi_create(None, 1, 8, None) == P.ERR_ARG
    rc = lib.f3ds_multi_create(None, 0, 8, ctypes.byref(h))
    assert rc in (P.ERR_ARG, P.ERR_NO_DEVICE) and not h.value
    if P.device_count() == 0:
        assert lib.f3ds_multi_create(None, 1, 8, ctypes.byref(h)) == P.ERR_NO_DEVICE and not h.value
        with pytest.raises(P.F3dsError) as e:
            P.MultiGpu(n_devices=1)
        assert e.value.code == P.ERR_NO_DEVICE
    assert lib.f3ds_multi_devices(None) == 0 and lib.f3ds_multi_device_of_frame(None, 3) == -1
    prm = P.default_params()
    assert lib.f3ds_multi_segment(None, None, None, 0, ctypes.byref(prm), None, None) == P.ERR_ARG
    assert lib.f3ds_multi_gathered_labels(None) is None
    t = ctypes.c_int(7)
    assert lib.f3ds_multi_submit(None, None, None, 0, ctypes.byref(prm), None, None, ctypes.byref(t)) == P.ERR_ARG      # the pipelined form
    assert lib.f3ds_multi_collect(None, 0) == P.ERR_ARG and lib.f3ds_multi_reserve(None, 1000) == P.ERR_ARG
    lib.f3ds_multi_destroy(None)
    assert isinstance(lib.f3ds_multi_last_error(), bytes)


def test_cpp_clustering_adapter(P, tmp_path):
    """include/f3ds_clustering.hpp -- the header-only C++ mirror of class Clustering (clustering.h:116-211) and of the
    pcl::SupervoxelClustering call sequence over the C-ABI -- compiles against libf3ds, throws the reference's exceptions before an
    initial state is set, and (on a GPU box) reproduces the labels of a plain f3ds_segment call."""
    src = tmp_path / "adapter.cpp"
    src.write_text(r'''
#include <cstdio>
#include <cstring>
#include <vector>
#include "f3ds_clustering.hpp"
int main() {
    f3ds::Clustering c;                                                    // defaults of clustering.cpp:533-539
    if (c.get_delta_c() != f3ds::LAB_CIEDE00 || c.get_delta_g() != f3ds::NORMALS_DIFF || c.get_merging() != f3ds::ADAPTIVE_LAMBDA) return 10;
    if (c.get_lambda() != 0.5f || c.get_bins_num() != 500) return 11;
    try { c.set_lambda(0.3f); return 12; } catch (const std::logic_error&) {}            // :574-582
    try { c.set_bins_num(10); return 13; } catch (const std::logic_error&) {}            // :589-597
    c.set_merging(f3ds::MANUAL_LAMBDA);
    try { c.set_lambda(1.5f); return 14; } catch (const std::invalid_argument&) {}
    c.set_lambda(0.25f); if (c.get_lambda() != 0.25f) return 15;
    c.set_merging(f3ds::EQUALIZATION);
    try { c.set_bins_num(-1); return 16; } catch (const std::invalid_argument&) {}
    try { c.cluster(0.2f); return 17; } catch (const std::logic_error&) {}               // :670-673
    {   // the two statics (clustering.h:207-210, clustering.cpp:793-846): label -> lookup-table colour, colour -> label in order of first appearance
        f3ds::LabeledCloud lc; lc.xyz = {0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4}; lc.label = {7, 3, 7, 260, 3};
        const f3ds::ColoredCloud cc = f3ds::Clustering::label2color(lc);
        if (cc.xyz != lc.xyz || cc.rgba.size() != 5) return 18;
        for (size_t i = 0; i < 5; ++i) if (cc.rgba[i] != (0xFF000000u | f3ds_label_color(lc.label[i]))) return 18;
        if (cc.rgba[3] != (0xFF000000u | f3ds_label_color(4))) return 18;                 // the table has 256 entries: label 260 wraps
        const f3ds::LabeledCloud back = f3ds::Clustering::color2label(cc);
        const std::vector<uint32_t> want = {0, 1, 0, 2, 1};
        if (back.xyz != lc.xyz || back.label != want) return 19;
    }
    if (f3ds_device_count() < 1) { std::puts("no device: surface only"); return 0; }
    const uint32_t W = 160, H = 120;
    std::vector<f3ds::PointXYZRGBA> pts((size_t)W * H);
    if (f3ds_synth_frame(0, 7, W, H, 30, pts.data())) return 20;
    f3ds::SupervoxelClustering super(0.02f, 0.2f);
    super.setInputCloud(pts.data(), pts.size());
    f3ds::Supervoxels sv = super.extract();
    f3ds::Clustering seg(f3ds::LAB_CIEDE00, f3ds::CONVEX_NORMALS_DIFF, f3ds::ADAPTIVE_LAMBDA);
    seg.set_initialstate(super);
    seg.cluster(0.2f);
    // the same frame through the plain C call
    f3ds_ctx* ctx; if (f3ds_create(0, &ctx)) return 21;
    f3ds_params p; f3ds_default_params(&p); p.voxel_res = 0.02f; p.seed_res = 0.2f; p.geom_metric = F3DS_CONVEX_NORMALS_DIFF; p.threshold = 0.2f;
    std::vector<uint32_t> want(pts.size()); f3ds_result r;
    if (f3ds_segment(ctx, pts.data(), pts.size(), 0, &p, want.data(), 0, &r)) return 22;
    if (seg.get_point_labels() != want) return 23;
    if (seg.result().n_regions != r.n_regions || sv.label.size() != r.n_supervoxels || seg.get_lambda() != r.lambda) return 24;
    f3ds::LabeledCloud lc = seg.get_labeled_cloud(); f3ds::ColoredCloud cc = seg.get_colored_cloud();
    if (lc.label.empty() || lc.label.size() != cc.rgba.size() || cc.rgba[0] != f3ds_label_color(lc.label[0])) return 25;
    if (super.getSupervoxelAdjacency().size() != r.n_edges) return 26;
    f3ds_destroy(ctx);
    std::puts("adapter ok");
    return 0;
}
''')
    pkg_dir = os.path.join(ROOT, "fast-3d-pointcloud-segmentation_amd")
    exe = tmp_path / "adapter"
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src), "-L", pkg_dir, "-lf3ds", "-Wl,-rpath," + pkg_dir], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert ("adapter ok" in r.stdout) == (P.device_count() > 0)


def test_no_register_copies_inside_hand_issued_lds_pipelines():
    """The fold loops of the merge kernel and the ordered sums of the voxel-normal kernel issue their LDS reads and the matching s_waitcnt by hand (asm volatile).
    Between a read and its wait the destination registers are not valid yet, which the compiler cannot know: anything it places there that touches them -- a copy, a
    spill, any VALU / DS use -- reads stale data, and a scalar load it schedules there makes a PARTIAL wait prove nothing (SMEM returns out of order on the same
    counter).  Round 4 hit the first case once (results that depended on timing).  tools/check_async_copies.py replays the LGKM counter over the generated gfx950
    assembly and reports every such instruction; hipcc cross-compiles without a GPU, so this runs -- and must run, not skip -- on the build box."""
    assert os.path.exists("/opt/rocm/bin/hipcc"), "hipcc is part of the image: this guard must not be skipped"
    r = subprocess.run([os.path.join(ROOT, "tools", "check_async_copies.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("0 suspicious copies") >= 6, r.stdout


def test_the_pipeline_guard_reports_what_it_is_there_for():
    """The checker is not vacuous: on hand-made listings it reports a register copy, a spill and a VALU use of a pending register, a scalar load ahead of a
    partial wait, and stays silent when the partial wait has retired the read (LDS returns in order) or a flat load has been waited for through vmcnt."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_async_copies", os.path.join(ROOT, "tools", "check_async_copies.py"))
    chk = importlib.util.module_from_spec(spec); spec.loader.exec_module(chk)
    issue = ["\t;;#ASMSTART", "\tds_read2_b32 v[10:11], v3 offset1:12", "\tds_read2_b32 v[12:13], v3 offset0:24 offset1:36", "\t;;#ASMEND"]
    wait = lambda n: ["\t;;#ASMSTART", "\ts_waitcnt lgkmcnt(%d)" % n, "\t;;#ASMEND"]
    kinds = lambda body: [k for _, k, _ in chk.scan(body)]
    assert kinds(issue + ["\tv_mov_b32_e32 v20, v11"] + wait(0)) == ["use"]
    assert kinds(issue + ["\tscratch_store_dword off, v12, s32 offset:16"] + wait(0)) == ["use"]
    assert kinds(issue + ["\tv_add_f32_e32 v13, v1, v2"] + wait(0)) == ["use"]                      # (overwriting a pending destination is a bug too)
    assert kinds(issue + wait(1) + ["\tv_add_f32_e32 v1, v10, v11"]) == []                         # first read retired by the partial wait
    assert kinds(issue + wait(1) + ["\tv_add_f32_e32 v1, v12, v11"]) == ["use"]                    # ... the second one not
    assert kinds(issue + ["\ts_load_dwordx2 s[4:5], s[0:1], 0x10"] + wait(1) + ["\tv_add_f32_e32 v1, v10, v11"]) == ["smem", "use"]
    assert kinds(["\ts_load_dwordx2 s[4:5], s[0:1], 0x10", "\ts_waitcnt lgkmcnt(0)"] + issue + wait(1) + ["\tv_add_f32_e32 v1, v10, v11"]) == []
    assert kinds(["\tflat_load_dword v1, v[4:5]", "\ts_waitcnt vmcnt(0)"] + issue + wait(1) + ["\tv_add_f32_e32 v1, v10, v11"]) == []
    assert kinds(["\tflat_load_dword v1, v[4:5]"] + issue + wait(1)) == ["smem"]
    assert kinds(issue + ["\tds_read_b32 v40, v41", "\ts_waitcnt lgkmcnt(1)", "\tv_mov_b32_e32 v1, v13"]) == []      # a compiler-issued LDS read after ours only strengthens the wait


def test_development_switches_are_gated_behind_f3ds_dev(P, monkeypatch):
    """the ~25 result-neutral F3DS_* switches are read only while F3DS_DEV is set (csrc/f3ds_dev.h).  Host side of it:
    the gate as the library sees it, and the version string that bench.py voids its value on."""
    lib = P.load_library()
    lib.f3ds_dev_mode.restype = ctypes.c_int
    monkeypatch.delenv("F3DS_DEV", raising=False)
    monkeypatch.setenv("F3DS_MERGE_NW", "8")            # a stray switch alone opens nothing
    assert lib.f3ds_dev_mode() == 0
    assert "+dev" not in lib.f3ds_version_string().decode()
    monkeypatch.setenv("F3DS_DEV", "0")
    assert lib.f3ds_dev_mode() == 0
    monkeypatch.setenv("F3DS_DEV", "1")
    assert lib.f3ds_dev_mode() == 1
    text = lib.f3ds_version_string().decode()
    assert text.endswith(" +dev") and P.library_stamp(lib)[0] == P.source_stamp()      # (the stamp still parses with the suffix)


def test_python_label2color_color2label(P):
    """The Python mirror of Clustering::label2color / color2label (clustering.cpp:793-846) agrees with the C++ one's definition."""
    xyz = np.arange(15, dtype=np.float32).reshape(5, 3)
    lab = np.array([7, 3, 7, 260, 3], np.uint32)
    x2, rgba = P.Clustering.label2color(xyz, lab)
    assert np.array_equal(x2, xyz) and list(rgba) == [0xFF000000 | P.label_color(int(l)) for l in lab]
    x3, back = P.Clustering.color2label(x2, rgba)
    assert np.array_equal(x3, xyz) and list(back) == [0, 1, 0, 2, 1]
