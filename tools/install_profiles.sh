#!/bin/bash
# Here, after `gpurun -- bash tools/refresh_profiles.sh <tag>`: copy the judged summaries from gpurun_out/ into profiles/ (the bench files keep their JSON line only).
tag=${1:-r6}
cd "$(dirname "$0")/.." || exit 1
for f in kernel_stats.csv kernel_stats_isolated.csv timeline.txt sweep_trace.txt pmc_hbm_traffic.json sq_counters.txt mem_pipeline_counters.txt config4_kernel_stats.csv config4_pmc_hbm_traffic.json config4_stages.txt; do cp gpurun_out/${tag}_$f profiles/; done
for f in bench.json bench_driver_flags.json bench_profiled_run.json; do tail -1 gpurun_out/${tag}_$f > profiles/${tag}_$f; done
for p in merge_prof config4_merge_prof; do [ -f gpurun_out/${tag}_${p}_raw.txt ] && grep -a "whole loop\|cycles/merge\|argmin phase\|record:\|speculation\|rows class\|touched class" gpurun_out/${tag}_${p}_raw.txt | tail -14 > profiles/${tag}_$p.txt; done
python3 - "$tag" <<'PY'
import json, csv, re, sys
tag = sys.argv[1]
for f in ("bench_driver_flags.json", "bench.json", "bench_profiled_run.json"):
    d = json.loads(open("profiles/%s_%s" % (tag, f)).read()); r = d["roofline"]
    print(f, "value", d["value"], "survey 8d", d["value_survey_8d"], "ms/step", d["ms_per_step"], "lone", d["single_frame_latency_ms"], "launch", r["launch_ms"], "frac", r["frac"], "whole", r["whole_path"]["frac"],
          "traffic", r["whole_path"]["traffic_per_frame"], r["whole_path"]["traffic_per_frame_min"], "cpu", d["cpu_baseline"] and d["cpu_baseline"]["value"], d["library"], "mismatches", len(d["labels_checked"]["mismatches"]))
out = {}; tot = 0.0
for r in csv.DictReader(open("profiles/%s_kernel_stats_isolated.csv" % tag)):
    m = re.search(r"d_[A-Za-z_0-9]+", r["Name"]); name = m.group(0) if m else r["Name"][:30]
    us = float(r["TotalDurationNs"]) / 1e3 / 1152; out[name] = out.get(name, 0) + us; tot += us
print({k: round(v, 1) for k, v in sorted(out.items(), key=lambda kv: -kv[1])[:12]}, "us per frame, total", round(tot, 1))
print(open("profiles/%s_timeline.txt" % tag).readline().strip())
PY
grep "^scene" profiles/${tag}_config4_stages.txt | cut -c1-60
