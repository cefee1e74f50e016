#!/usr/bin/env python3
"""Per-sweep durations (us) of the sweep kernels of the last batched dispatch sequence in a rocprofv3 kernel trace.
usage: tools/sweep_trace.py <dir> <frames per batch>"""
import csv, glob, re, collections, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
nf = sys.argv[2]
seq = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    m = re.search(r"d_([A-Za-z_0-9]+)", r["Kernel_Name"])
    if not m or r["Grid_Size_Y"] != nf:
        continue
    seq[m.group(0)].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
for n in ("d_sweep_begin", "d_sweep_R_first", "d_sweep_R_round", "d_sweep_R_pre", "d_sweep_R", "d_sweep_R_tail", "d_sweep_claim", "d_claim_mark", "d_centroid", "d_centroid_mark"):
    v = sorted(seq.get(n, []))
    if not v:
        continue
    # (sweeps 0 and 1 record no incremental launches; since the end of round 4 round 0 shares a dispatch with the pre-pass -- d_sweep_R_first -- and the marking passes
    # d_claim_mark / d_centroid_mark are part of d_sweep_claim / d_centroid: older traces still have them)
    per = {"d_sweep_R_round": 42 if "d_sweep_R_first" not in seq else 28, "d_claim_mark": 15, "d_centroid_mark": 15}.get(n, 16)
    last = [round(x[1]) for x in v[-per:]]
    print("%-16s sum %6d us  %s" % (n, sum(last), last))
