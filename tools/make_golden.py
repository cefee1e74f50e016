#!/usr/bin/env python3
"""Regenerate tests/golden/oracle_golden.json from the CPU oracle (run in the build container).

For every case: the scalar summary, a SHA-256 of every intermediate array the oracle exposes, and
-- for the small case -- the merge sequence and lambda in clear.  The -m "not gpu" suite checks the
oracle (and the device emulation) against this file; the -m gpu suite checks the HIP path against it
too.  Nothing here comes from the reference's sources: inputs are synthetic frames / the reference's
bundled PCD, outputs are the oracle's."""
import hashlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import sha_of, ALL_DEBUG, CpuChecker, FIXTURE_PCD, pkg
from golden_cases import GOLDEN_CASES, case_points, case_params

P = pkg()
ora = CpuChecker(os.path.join(ROOT, "oracle", "libf3ds_oracle.so"), "f3ds_oracle")
out = {}
for name in GOLDEN_CASES:
    pts = case_points(P, name)
    prm = case_params(P, name)
    rc, labels, res, h = ora.segment(pts, prm)
    assert rc == 0, (name, rc)
    entry = {"summary": {k: (v if not isinstance(v, float) or v == v else "nan") for k, v in res.as_dict().items() if k not in ("ms_stage", "ms_total")},
             "sha256": {w: sha_of(h.get(w)) for w in ALL_DEBUG},
             "labels_sha256": sha_of(labels),
             "label_histogram": np.bincount(labels[labels != 0xFFFFFFFF]).tolist()[:64],
             "unlabelled_points": int((labels == 0xFFFFFFFF).sum())}
    if res.n_merges <= 400:
        entry["merges"] = h.get("MERGES").reshape(-1, 3).tolist()
    out[name] = entry
    print(name, entry["summary"])
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "oracle_golden.json"), "w"), indent=1, sort_keys=True)
