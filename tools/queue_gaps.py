#!/usr/bin/env python3
"""Per HSA queue (= host thread of bench.py) in a rocprofv3 kernel trace: time inside kernels vs gaps between them,
and how much longer each kernel takes than its shortest instance (sharing the chip).
usage: tools/queue_gaps.py <dir with *kernel_trace.csv> [min grid.y]"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
miny = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    m = re.search(r"d_([A-Za-z_0-9]+)", r["Kernel_Name"])
    rows[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0) if m else r["Kernel_Name"][:30], int(r["Grid_Size_Y"])))
for q, v in sorted(rows.items()):
    v.sort()
    v = [x for x in v if x[3] >= miny]
    if len(v) < 100:
        continue
    span = v[-1][1] - v[0][0]
    busy = 0; gaps = collections.Counter(); last_end = v[0][0]; where = collections.Counter(); prev = "-"
    for s, e, n, gy in v:
        if s > last_end:
            g = s - last_end
            gaps["<20us" if g < 20e3 else ("<200us" if g < 200e3 else ("<2ms" if g < 2e6 else ">=2ms"))] += g
            if g >= 0.5e6:
                where[prev + " -> " + n] += g
        busy += max(0, e - max(s, last_end)); last_end = max(last_end, e); prev = n
    print("queue %s: %d launches over %.0f ms; inside kernels %.0f ms (%.0f%%); gaps by length (ms): %s" % (
        q, len(v), span / 1e6, busy / 1e6, 100 * busy / span, {k: round(x / 1e6, 1) for k, x in sorted(gaps.items())}))
    for k, x in where.most_common(8):
        print("    %-50s %7.1f ms" % (k, x / 1e6))
