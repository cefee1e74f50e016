#!/usr/bin/env python3
"""Bytes per FETCH_SIZE / WRITE_SIZE unit for known access patterns (tools/ubench/ubench_fetch.hip), from two rocprofv3
--pmc passes.  usage: tools/pmc_calibrate.py <fetch_dir> <write_dir> <out.json>"""
import collections, csv, glob, json, sys
GiB = 1 << 30
known = {"k_stream16": ("read", 4 * GiB), "k_stream4": ("read", 1 * GiB), "k_gather4": ("read_lines", 32 << 20), "k_gather16": ("read_lines", 32 << 20),
         "k_write16": ("write", 4 * GiB), "k_write4": ("write", 1 * GiB), "k_scatter4": ("write_lines", 32 << 20)}
def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            for k in known:
                if k + "(" in r["Kernel_Name"] or r["Kernel_Name"].startswith(k):
                    agg[k].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k, (kind, amount) in known.items():
    c = fe.get(k) if kind.startswith("read") else wr.get(k)
    other = wr.get(k) if kind.startswith("read") else fe.get(k)
    if not c:
        continue
    if kind.endswith("lines"):
        out[k] = {"counter_KB": c, "lines_touched": amount, "bytes_per_line_if_KB_is_1024": round(c * 1024 / amount, 2), "other_counter_KB": other}
    else:
        out[k] = {"counter_KB": c, "bytes_moved": amount, "true_bytes_per_counter_KB": round(amount / c, 1), "factor_vs_1024": round(amount / c / 1024, 3), "other_counter_KB": other}
    print(k, out[k])
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
