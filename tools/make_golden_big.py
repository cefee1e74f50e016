#!/usr/bin/env python3
"""Regenerate tests/golden/oracle_golden_big.json from the CPU oracle (run in the build container, ~2 min on 8 cores).

BASELINE.json's full sizes, pinned by hashes instead of re-running the oracle on the GPU box:
  config5  -- the 64 synthetic 1M-point frames of the batch (seeds 1000..1063, flags -v 0.008 -s 0.08 --AL --CVX -t 0.2):
              SHA-256 of the per-point labels and of the merge sequence, scalar summary, per frame;
  config4  -- the 20M-point fused scene (-v 0.02 -s 0.2 --NT): SHA-256 of labels, MERGES, VOXEL_KEYS, EDGES, summary.
Inputs are libf3ds' deterministic generators, outputs the oracle's; nothing comes from the reference's sources."""
import hashlib, json, os, sys
from multiprocessing import Pool
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

SUMMARY = ("n_points", "n_finite", "n_voxels", "octree_depth", "n_seed_cells", "n_seeds", "n_supervoxels", "n_edges", "n_merges", "n_regions", "sweeps")


def run(job):
    from conftest import sha_of, CpuChecker, pkg
    P = pkg()
    ora = CpuChecker(os.path.join(ROOT, "oracle", "libf3ds_oracle.so"), "f3ds_oracle")
    name, synth, over, arrays = job
    pts = P.synth_frame(*synth)
    prm = P.launch_params(**over)
    rc, labels, res, h = ora.segment(pts, prm)
    assert rc == 0, (name, rc)
    e = {"synth": list(synth), "params": over, "summary": {k: getattr(res, k) for k in SUMMARY},
         "labels_sha256": sha_of(labels),
         "sha256": {w: sha_of(h.get(w)) for w in arrays}}
    h.close()
    print(name, e["summary"], flush=True)
    return name, e


# an ORGANISED frame beyond 1M points (2 560 x 2 048 = 5.2M, -v 0.006 -s 0.06: at most ~1 000 distinct voxels per tile of 4 096 points and 168 points per voxel, inside what the tile path takes): stage 0's tile path at five times the bench frame's size
ORGANISED_5M = ("organised_5m_frame", (0, 4242, 2560, 2048, 30), dict(voxel_res=0.006, seed_res=0.06),
                ("GRID", "VOXEL_KEYS", "VOXEL_COUNT", "VOXEL_XYZ", "VOXEL_RGB", "VOXEL_NORMAL", "POINT_VOXEL", "VOXEL_SVLABEL", "MERGES"))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--only-organised":      # (adds / refreshes that one entry: ~40 s instead of the whole file's minutes)
        path = os.path.join(ROOT, "tests", "golden", "oracle_golden_big.json")
        out = json.load(open(path)); out.update([run(ORGANISED_5M)])
        json.dump(out, open(path, "w"), indent=1, sort_keys=True)
        sys.exit(0)
    jobs = [("config5_seed%d" % s, (0, s, 1000, 1000, 30), dict(voxel_res=0.008, seed_res=0.08), ("MERGES", "VOXEL_SVLABEL")) for s in range(1000, 1064)]
    with Pool(8) as pool:
        out = dict(pool.map(run, jobs, chunksize=1))
    out.update([run(("config4_20m_scene", (1, 3000, 5000, 4000, 0), dict(voxel_res=0.02, seed_res=0.2, use_transform=0), ("MERGES", "VOXEL_KEYS", "EDGES", "VOXEL_SVLABEL")))])
    out.update([run(ORGANISED_5M)])
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "oracle_golden_big.json"), "w"), indent=1, sort_keys=True)
