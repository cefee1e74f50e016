#!/usr/bin/env python3
"""Guard for the hand-issued LDS pipelines (the fold loops of d_merge_il_t, the ordered sums of d_normals_t).

Between an `asm volatile` ds_read and the hand-written `s_waitcnt lgkmcnt(n)` that covers it the destination registers hold nothing yet, which
the compiler cannot know: the asm statements declare them as ordinary outputs.  Round 4 hit the consequence once (register copies placed between a
read and its wait: labels that depended on timing).  This script compiles the device code to gfx950 assembly and replays the LGKM counter over every
function whose name contains one of the given substrings:

  * every LDS operation enters a queue in program order (LDS returns in order); a hand-issued ds_read also records its destination registers;
  * `s_waitcnt lgkmcnt(k)` -- hand-written or inserted by the compiler -- retires all but the youngest k queue entries; lgkmcnt(0) and s_barrier
    (always preceded by a full wait) retire everything;
  * scalar memory loads and flat loads share the counter but return OUT of order: while one is outstanding a partial wait proves nothing about the
    LDS reads before it.  A hand-written partial wait met in that state is reported ("smem"), and retires nothing.  (A flat access leaves the
    counter once `s_waitcnt vmcnt(0)` has seen its data arrive; a scalar load only at lgkmcnt(0).);
  * ANY instruction -- VALU, DS, VMEM, scratch spill, hand-written or not -- that reads or writes a register of a not-yet-retired hand-issued read
    is reported ("use").

Straight-line replay (branches are not followed): a report is a reason to read the listing, not a proof of a bug; zero reports over the pipelines'
code is what the CPU test suite asserts.  A pipeline added to the kernels needs no change here as long as it is issued from inline asm.
usage: tools/check_async_copies.py [kernel-name substring ...]   (default: d_merge_il_t d_normals_t); exit code 1 when something is reported."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fast-3d-pointcloud-segmentation_amd", "csrc")
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out |= set(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def scan(lines):
    queue = []            # LGKM operations in flight, oldest first: (kind, registers) with kind in {"hand", "lds", "ooo"}
    found, inasm = [], False

    def pending():
        p = set()
        for kind, r in queue:
            if kind == "hand":
                p |= r
        return p

    for n, l in enumerate(lines):
        t = l.split(";")[0].strip() if not l.strip().startswith(";;#ASM") else l.strip()
        if t.startswith(";;#ASMSTART"):
            inasm = True; continue
        if t.startswith(";;#ASMEND"):
            inasm = False; continue
        if not t or t.endswith(":") or t.startswith("."):
            continue
        op = t.split()[0]
        args = t[len(op):]
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                k = int(m.group(1))
                ooo = any(kind in ("ooo", "flat") for kind, _ in queue)
                if k == 0:
                    queue = []
                elif ooo:
                    if inasm and pending():
                        found.append((n + 1, "smem", t))
                else:
                    queue = queue[len(queue) - k:] if k < len(queue) else queue
            elif "lgkmcnt" not in t and re.fullmatch(r"s_waitcnt\s+(0|0x0)", t):
                queue = []
            if re.search(r"vmcnt\(0\)", t):      # a flat load that has delivered its data (vmcnt) has left the LGKM counter as well
                queue = [e for e in queue if e[0] != "flat"]
            continue
        if op == "s_barrier":
            queue = []; continue
        p = pending()
        if p:
            hit = vregs(args) & p
            if hit and not (inasm and op.startswith("ds_read")):      # (a hand-issued read re-using a register of an un-retired one would also be a bug, but its own operand list is the issue below)
                found.append((n + 1, "use", t))
        if op.startswith("ds_"):
            if inasm and op.startswith("ds_read"):
                dst = args.split(",")[0]
                hit = vregs(dst) & p
                if hit:
                    found.append((n + 1, "use", t))
                queue.append(("hand", vregs(dst)))
            else:
                queue.append(("lds", set()))
        elif op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memtime") or op.startswith("s_memrealtime"):
            queue.append(("ooo", set()))
        elif op.startswith("flat_"):
            queue.append(("flat", set()))
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
                for n, kind, t in f[:12]:
                    print("    line %d of the function [%s]: %s" % (n, kind, t))
                bad += len(f)
            name = None
        elif name:
            body.append(l)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
