#!/usr/bin/env python3
"""Where the chip's instruction-issue capacity goes: per kernel, summed over every launch of a rocprofv3 --pmc pass
(SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAVES), per frame.
SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES count quad-cycles per wave (guides/MI355X_MICROARCH.md): x 4 = SIMD cycles.
usage: tools/pmc_issue.py <dir> <frames in the run> [clock GHz]"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
frames = float(sys.argv[2]); ghz = float(sys.argv[3]) if len(sys.argv) > 3 else 2.4
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    m = re.search(r"d_([A-Za-z_0-9]+)", r["Kernel_Name"])
    name = re.sub(r"_tILi.*", "_t", m.group(0)) if m else r["Kernel_Name"][:30]
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
SIMDS = 1024
tot = collections.defaultdict(float)
rows = []
for k, c in agg.items():
    valu = 4 * c.get("SQ_ACTIVE_INST_VALU", 0) / frames; lds = 4 * c.get("SQ_ACTIVE_INST_LDS", 0) / frames; anyi = 4 * c.get("SQ_ACTIVE_INST_ANY", 0) / frames
    wc = 4 * c.get("SQ_WAVE_CYCLES", 0) / frames
    rows.append((k, valu, lds, anyi, wc, c.get("SQ_INSTS_VALU", 0) / frames, c.get("SQ_WAVES", 0) / frames))
    for i, v in enumerate((valu, lds, anyi, wc)):
        tot[i] += v
print("per frame; 'chip us' = SIMD cycles / (%d SIMDs x %.1f GHz)" % (SIMDS, ghz))
print("%-22s %14s %10s %14s %10s %14s %12s %10s" % ("kernel", "VALU busy cyc", "chip us", "LDS busy cyc", "chip us", "any-inst cyc", "wave cycles", "waves"))
for k, valu, lds, anyi, wc, nv, w in sorted(rows, key=lambda r: -r[1])[:24]:
    print("%-22s %14.0f %10.1f %14.0f %10.1f %14.0f %12.0f %10.0f" % (k, valu, valu / SIMDS / ghz / 1e3, lds, lds / SIMDS / ghz / 1e3, anyi, wc, w))
print("%-22s %14.0f %10.1f %14.0f %10.1f %14.0f %12.0f" % ("all", tot[0], tot[0] / SIMDS / ghz / 1e3, tot[1], tot[1] / SIMDS / ghz / 1e3, tot[2], tot[3]))
