#!/usr/bin/env python3
"""Per-frame kernel time table from a rocprofv3 --kernel-trace --stats csv (batched launches).
usage: tools/kstats.py <dir with *kernel_stats.csv> <frames processed in the run>"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
frames = float(sys.argv[2])
rows = []
for r in csv.DictReader(open(f)):
    m = re.search(r"d_([A-Za-z_0-9]+)", r["Name"])
    rows.append((m.group(0) if m else r["Name"][:40], int(r["Calls"]), float(r["TotalDurationNs"])))
tot = sum(r[2] for r in rows)
print("%-28s %8s %12s %10s" % ("kernel", "calls", "us/frame", "share"))
for n, c, t in sorted(rows, key=lambda r: -r[2]):
    print("%-28s %8d %12.1f %9.1f%%" % (n, c, t / frames / 1e3, 100 * t / tot))
print("%-28s %8s %12.1f" % ("sum", "", tot / frames / 1e3))
