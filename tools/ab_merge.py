#!/usr/bin/env python3
"""A/B of builds of libf3ds.so on the merge stage of one 1M-point frame: alternating subprocess runs, median and minimum
of the stage's device time.  usage: tools/ab_merge.py <lib A> <lib B> ... [rounds]   (clock / DVFS noise on one box is ~ +-1 ms: single runs mislead)"""
import os, subprocess, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import importlib, os, sys
sys.path.insert(0, %r)
import torch
P = importlib.import_module("fast-3d-pointcloud-segmentation_amd")
prm = P.launch_params(voxel_res=0.008, seed_res=0.08)
f = P.synth_frame(0, 1000, 1000, 1000, 30); c = P.Context(0)
ts = []
for r in range(8):
    c.segment(f, prm); ts.append(c.result.ms_stage[5])
print(" ".join("%%.3f" %% t for t in ts[2:]))
''' % ROOT
libs = [a for a in sys.argv[1:] if not a.isdigit()]; rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 4
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, F3DS_LIB=l, F3DS_DEV="1")).stdout.strip().splitlines()[-1]
        res[l] += [float(x) for x in out.split()]
for l in libs:
    v = res[l]
    print("%-60s merge stage: median %.2f ms, min %.2f, max %.2f  (%d runs)" % (os.path.basename(l), statistics.median(v), min(v), max(v), len(v)))
