#!/usr/bin/env python3
"""What the GPU does during the timed region of bench.py, from a rocprofv3 --kernel-trace csv: wall span, time with at least
one kernel running, average number of kernels running, and per kernel its summed duration, launches and mean duration.
The timed region is taken as the last `--tail-frac` of the trace's launches with grid.y >= 2 (the batched dispatches).
usage: tools/timeline.py <dir with *kernel_trace.csv> [tail fraction, default 0.4]"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
ev = []
for r in csv.DictReader(open(f)):
    m = re.search(r"d_([A-Za-z_0-9]+)", r["Kernel_Name"])
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0) if m else r["Kernel_Name"][:30], int(r["Grid_Size_Y"]) // max(1, int(r.get("Workgroup_Size_Y", 1) or 1)), int(r["Grid_Size_X"]), r["Queue_Id"]))
ev.sort()
t_end = ev[-1][1]; t_begin = ev[0][0]
cut = t_end - (t_end - t_begin) * frac
ev = [e for e in ev if e[0] >= cut]
span = ev[-1][1] - ev[0][0]
pts = []
for s, e, *_ in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
busy = 0; area = 0; cur = 0; last = pts[0][0]; hist = collections.Counter()
for t, d in pts:
    if cur > 0:
        busy += t - last
    area += cur * (t - last); hist[min(cur, 8)] += t - last
    cur += d; last = t
print("window %.1f ms: some kernel running %.1f%% of it, %.2f kernels running on average; time with k kernels running: %s" % (
    span / 1e6, 100.0 * busy / span, area / span, {k: "%.0f%%" % (100.0 * v / span) for k, v in sorted(hist.items())}))
per = collections.defaultdict(lambda: [0, 0])
for s, e, n, gy, gx, q in ev:
    per[n][0] += e - s; per[n][1] += 1
tot = sum(v[0] for v in per.values())
print("%-26s %8s %12s %12s %8s" % ("kernel", "launches", "sum ms", "mean us", "share"))
for n, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:28]:
    print("%-26s %8d %12.1f %12.1f %7.1f%%" % (n, c, t / 1e6, t / c / 1e3, 100.0 * t / tot))
print("%-26s %8d %12.1f" % ("all", sum(v[1] for v in per.values()), tot / 1e6))
