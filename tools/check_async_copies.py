#!/usr/bin/env python3
"""Guard for the hand-issued LDS pipelines (the fold loops of d_merge_il_t, the ordered sums of d_normals_t): between an `asm volatile` ds_read and the
hand-written s_waitcnt that covers it, the destination registers hold nothing yet -- but the compiler does not know that, and may place a register copy there
(it did: the arms of an if / else around two read sets were unified with v_mov copies BEFORE one arm's wait; results then depended on timing).  This script
compiles the device code to assembly and reports every v_mov whose source is a register that a hand-written ds_read has requested and no lgkmcnt(0) / barrier has
covered yet.  Conservative (partial waits do not clear anything): a report is a reason to read the listing, not a proof of a bug.
usage: tools/check_async_copies.py [kernel-name substring ...]   (default: d_merge_il_t d_normals_t); exit code 1 when something is reported."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fast-3d-pointcloud-segmentation_amd", "csrc")


def regs(spec):
    m = re.match(r"v\[(\d+):(\d+)\]", spec)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)", spec)
    return {int(m.group(1))} if m else set()


def scan(lines):
    pending, found, inasm = set(), [], False
    for n, l in enumerate(lines):
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            inasm = True; continue
        if t.startswith(";;#ASMEND"):
            inasm = False; continue
        if inasm:
            if t.startswith("ds_read"):
                pending |= regs(t.split()[1].rstrip(","))
            if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                pending = set()
            continue
        if (t.startswith("s_waitcnt") and "lgkmcnt(0)" in t) or t.startswith("s_barrier"):
            pending = set()
        m = re.match(r"(v_mov_b32_e32|v_mov_b64_e32|v_accvgpr_write_b32)\s+(\S+),\s*(\S+)", t)
        if m and pending and regs(m.group(3)) & pending:
            found.append((n + 1, t))
    return found


def main():
    want = sys.argv[1:] or ["d_merge_il_t", "d_normals_t"]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "dev.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", out,
                        os.path.join(CSRC, "f3ds_hip.hip")], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().split("\n")
    bad = 0
    name, body = None, []
    for l in text + [".Lfunc_end"]:
        m = re.match(r"^(_ZN\S+):\s", l)
        if m:
            name, body = m.group(1), []
        elif l.startswith(".Lfunc_end") and name:
            if any(w in name for w in want):
                f = scan(body)
                short = re.search(r"(d_[a-z_]+_tILi\d+E(?:Li\d+E)?)", name)
                print("%-28s %6d instructions, %d suspicious copies" % (short.group(1) if short else name[:28], len(body), len(f)))
                for n, t in f[:8]:
                    print("    line %d of the function: %s" % (n, t))
                bad += len(f)
            name = None
        elif name:
            body.append(l)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
