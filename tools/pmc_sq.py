#!/usr/bin/env python3
"""Per-kernel SQ counter table from one rocprofv3 --pmc pass (largest-grid launches only; averages per launch).
usage: tools/pmc_sq.py <dir with *counter_collection.csv> [kernel name filter ...]"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
want = sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
for r in csv.DictReader(open(f)):
    m = re.search(r"d_([A-Za-z_0-9]+)", r["Kernel_Name"])
    name = m.group(0) if m else r["Kernel_Name"][:30]
    if want and not any(w in name for w in want):
        continue
    agg[name][int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name in sorted(agg):
    g = max(agg[name])
    c = {k: sum(v) / len(v) for k, v in agg[name][g].items()}
    print(name, "grid", g, "launches", len(next(iter(agg[name][g].values()))))
    for k in sorted(c):
        print("   %-24s %14.0f" % (k, c[k]))
    w = c.get("SQ_WAVES", 0)
    if "SQ_LDS_IDX_ACTIVE" in c:
        # SQ_LDS_IDX_ACTIVE = LDS-array cycles, SQ_LDS_BANK_CONFLICT = the extra ones (guides/MI355X_MICROARCH.md, LDS); SQ_ACTIVE_INST_VALU counts quad-cycles
        print("   LDS: %.0f instructions / wave, %.1f array cycles each, %.0f%% of the array cycles are bank conflicts | VALU busy %.0f%% of wave cycles" % (
            c.get("SQ_INSTS_LDS", 0) / w, c["SQ_LDS_IDX_ACTIVE"] / max(c.get("SQ_INSTS_LDS", 0), 1), 100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c["SQ_LDS_IDX_ACTIVE"], 1),
            100 * c.get("SQ_ACTIVE_INST_VALU", 0) / max(c.get("SQ_WAVE_CYCLES", 0), 1)))
    if w and "SQ_WAIT_ANY" in c:
        print("   per wave: cycles %.0f (x4 quad) | wait_any %.0f%% wait_inst %.0f%% active %.0f%% | valu %.0f vmem_rd %.0f salu %.0f" % (
            4 * c["SQ_WAVE_CYCLES"] / w, 100 * c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], 100 * c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"],
            100 * c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], c.get("SQ_INSTS_VALU", 0) / w, c.get("SQ_INSTS_VMEM_RD", 0) / w, c.get("SQ_INSTS_SALU", 0) / w))
