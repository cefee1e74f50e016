#!/usr/bin/env python3
"""What the GPU does during the timed region of bench.py, from a rocprofv3 --kernel-trace csv: wall span, time with at least
one kernel running, average number of kernels running, and per kernel its summed duration, launches and mean duration.
The timed region: with `calls=N` it is found from the trace itself -- bench.py's timed calls are the last N launches of the 4-wave merge
kernel (d_merge_il_t<4, .>: the warm-up's calls come before them, the lone frames and one-call latency runs behind them take the 8-wave
kernel), so the window runs from the end of the launch before those N to the end of the last of them.  Otherwise the last `tail fraction`
of the trace is taken.
usage: tools/timeline.py <dir with *kernel_trace.csv> [tail fraction, default 0.4 | calls=N]"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
arg = sys.argv[2] if len(sys.argv) > 2 else "0.4"
ev = []
merges4 = []
for r in csv.DictReader(open(f)):
    m = re.search(r"d_([A-Za-z_0-9]+)", r["Kernel_Name"])
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0) if m else r["Kernel_Name"][:30], int(r["Grid_Size_Y"]) // max(1, int(r.get("Workgroup_Size_Y", 1) or 1)), int(r["Grid_Size_X"]), r["Queue_Id"]))
    if re.search(r"d_merge_il_t<4,", r["Kernel_Name"]):
        merges4.append((int(r["End_Timestamp"]), int(r["Start_Timestamp"])))
ev.sort()
t_end = ev[-1][1]; t_begin = ev[0][0]
if arg.startswith("calls="):
    n = int(arg[6:]); merges4.sort()
    if len(merges4) <= n:
        sys.exit("fewer than %d launches of the 4-wave merge kernel in the trace" % (n + 1))
    cut, t_end = merges4[-n - 1][0], merges4[-1][0]
    ev = [e for e in ev if e[0] >= cut and e[1] <= t_end]
    d = [(e - s) / 1e6 for e, s in merges4[-n:]]
    print("timed region = the last %d launches of d_merge_il_t<4,.>: mean %.2f ms per launch (min %.1f, max %.1f)" % (n, sum(d) / n, min(d), max(d)))
else:
    cut = t_end - (t_end - t_begin) * float(arg)
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
