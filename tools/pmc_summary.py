#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE are collected in SEPARATE runs, as
guides/MI355X_MICROARCH.md prescribes) into HBM bytes per launch and kernel, and for the whole path per frame.

Units, calibrated on this box with tools/ubench/ubench_fetch.hip + tools/pmc_calibrate.py (profiles/r2_pmc_calibration.json):
  FETCH_SIZE (KB)  coalesced streams, 4 or 16 bytes per lane: true bytes = KB x 2048 (128-byte requests tallied as 64);
                   isolated gathers (one line per lane): KB x 1024 = 64 bytes per line touched -- the counter is exact there.
                   A kernel that mixes both lies between KB x 1024 and KB x 2048; there is no 128-byte request counter on
                   gfx950 (TCC_BUBBLE reads 0) to split them.  So: read_bytes_max = KB x 2048 for every kernel, read_bytes_min
                   = KB x 2048 for the pure stream kernels (STREAM below) and KB x 1024 for the others.
  WRITE_SIZE (KB)  exact for streams (KB x 1024); scattered 4-byte stores count 32 bytes each.
usage: tools/pmc_summary.py <fetch_dir> <write_dir> <frames_per_batched_launch> <out.json> [launches per frame json]"""
import collections, csv, glob, json, re, sys
fetch_dir, write_dir, nf, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
STREAM = {"d_bbox", "d_keys", "d_radix_hist", "d_radix_scatter", "d_radix_scatter_k", "d_heads", "d_segstart", "d_scan_tiles", "d_scan_add", "d_scan_single",
          "d_fill_u32", "d_fill_f32", "d_copy_u32", "d_iota", "d_voxel_accum", "d_chunkbox", "d_edge_init", "d_seg_count", "d_seg_write", "d_tile_keys", "d_multi_op"}
def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"d_([A-Za-z_0-9]+)", r["Kernel_Name"])
        name = m.group(0) if m else r["Kernel_Name"][:40]
        name = re.sub(r"_tILi.*", "_t", name)
        agg[name][int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    return agg
fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
res = {}
path_min = path_max = 0.0
for k in sorted(fe):
    g = max(fe[k])                       # the batched launches have the largest grid
    fv = sum(fe[k][g]) / len(fe[k][g])
    wv = sum(wr[k][g]) / len(wr[k][g]) if k in wr and g in wr[k] else 0.0
    launches = sum(len(v) for v in fe[k].values())      # all launches of this kernel in the profiled run
    total_f = sum(sum(v) for v in fe[k].values()); total_w = sum(sum(v) for v in wr[k].values()) if k in wr else 0.0
    lo = 2048 if k in STREAM else 1024
    res[k] = {"launches_of_the_largest_grid": len(fe[k][g]), "grid_size": g, "fetch_KB_raw": round(fv, 1), "class": "stream" if k in STREAM else "mixed",
              "read_bytes_min": int(fv * lo), "read_bytes_max": int(fv * 2048), "write_bytes": int(wv * 1024),
              "hbm_bytes_per_launch": int(fv * 2048 + wv * 1024), "hbm_bytes_per_frame": int((fv * 2048 + wv * 1024) / nf),
              "run_total_read_KB_raw": round(total_f, 1), "run_total_write_KB": round(total_w, 1), "run_launches": launches}
    path_min += total_f * lo + total_w * 1024; path_max += total_f * 2048 + total_w * 1024
frames_in_run = float(sys.argv[5]) if len(sys.argv) > 5 else None
doc = {"frames_per_launch": nf, "note": "read bytes: FETCH_SIZE KB x 2048 (stream kernels; upper bound for the others) or x 1024 (lower bound for kernels that gather); write bytes = WRITE_SIZE KB x 1024; "
                                        "hbm_bytes_per_launch / per_frame use the upper bound", "kernels": res}
if frames_in_run:
    doc["frames_in_profiled_run"] = frames_in_run
    doc["whole_path_hbm_bytes_per_frame_min"] = int(path_min / frames_in_run); doc["whole_path_hbm_bytes_per_frame"] = int(path_max / frames_in_run)
json.dump(doc, open(out, "w"), indent=1, sort_keys=True)
if frames_in_run:
    print("whole path: %.0f - %.0f MB of HBM traffic per frame (all launches of the run / %d frames)" % (path_min / frames_in_run / 1e6, path_max / frames_in_run / 1e6, frames_in_run))
for k, v in sorted(res.items(), key=lambda kv: -(kv[1]["run_total_read_KB_raw"] * 2 + kv[1]["run_total_write_KB"]))[:16]:
    print("%-20s %-6s %9.1f MB/launch  %8.2f MB/frame/launch   run total %8.1f MB over %d launches" % (k, v["class"], v["hbm_bytes_per_launch"] / 1e6, v["hbm_bytes_per_frame"] / 1e6,
          (v["run_total_read_KB_raw"] * 2048 + v["run_total_write_KB"] * 1024) / 1e6, v["run_launches"]))
