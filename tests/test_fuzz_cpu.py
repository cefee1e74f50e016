"""Seeded random frames and parameters: the sequential emulation of the device algorithms (tests/emul, same headers as
the kernels) against the literal oracle on every intermediate array.  tools/fuzz_cpu.py runs the same sweep at length;
the GPU counterpart is test_gpu_parity.py::test_random_small_frames_and_parameters."""
import numpy as np

from conftest import ALL_DEBUG, same_bits


def test_emulation_matches_oracle_on_random_cases(P, oracle, emul):
    rng = np.random.default_rng(424242)
    checked = 0
    for it in range(16):
        w, hgt = int(rng.integers(20, 160)), int(rng.integers(20, 120))
        kind = int(rng.integers(0, 2))
        pts = P.synth_frame(kind, int(rng.integers(1, 10**6)), w, hgt, int(rng.integers(0, 400)) if kind == 0 else 0)
        vres = float(rng.choice([0.015, 0.02, 0.03, 0.05, 0.08]))
        prm = P.launch_params(voxel_res=vres, seed_res=vres * float(rng.choice([2, 3, 5, 8, 12])), use_transform=int(rng.integers(0, 2)) if kind == 0 else 0,
                              color_metric=int(rng.integers(0, 2)), geom_metric=int(rng.integers(0, 2)), merging=int(rng.integers(0, 3)),
                              lambda_=float(rng.uniform(0.0, 1.0)), bins=int(rng.choice([0, 5, 20, 100])), threshold=float(rng.choice([0.0, 0.1, 0.3, 1.0])),
                              leaf_order=int(rng.integers(0, 2)), w_color=float(rng.uniform(0.0, 1.0)), w_spatial=float(rng.uniform(0.0, 1.0)),
                              w_normal=float(rng.uniform(0.0, 6.0)))
        rc, olab, ores, oh = oracle.segment(pts, prm)
        rc2, elab, eres, eh = emul.segment(pts, prm)
        assert rc == rc2, (it, rc, rc2)
        if rc:
            continue
        assert np.array_equal(olab, elab), it
        for k in ALL_DEBUG:
            assert same_bits(oh.get(k), eh.get(k)), (it, k)      # (a NaN's sign / payload depends on the compiler's operand order: not compared)
        checked += 1
    assert checked >= 10


def test_emulation_matches_oracle_on_degenerate_clouds(oracle, emul):       # (the fixtures build the two libraries if needed)
    """tools/fuzz_clouds.py on the CPU: empty / one-point / duplicate / collinear / planar / NaN / inf / negative-z clouds."""
    import os, subprocess, sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_clouds.py"), "80", "9"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "mismatches 0" in r.stdout.splitlines()[-1], r.stdout[-2000:]
