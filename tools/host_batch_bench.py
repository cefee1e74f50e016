#!/usr/bin/env python3
"""f3ds_segment_batch with host (pinned) buffers on either side, 3 host threads x 192 frames like bench.py:
which direction of the PCIe traffic costs what."""
import argparse, ctypes, importlib, os, sys, threading, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = importlib.import_module("fast-3d-pointcloud-segmentation_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=192); ap.add_argument("--groups", type=int, default=3); ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--variants", default="11,01,10,00", help="points on device? labels on device? per variant")
ap.add_argument("--stagger-ms", type=float, default=0.0, help="start group g that many ms x g late")
a = ap.parse_args()
lib = P.load_library(); prm = P.launch_params(); npts = 1000 * 1000; dev = torch.device("cuda", 0)
src = [P.synth_frame(0, 1000 + i, 1000, 1000, 30) for i in range(4)]
vp = ctypes.c_void_p
for pin, lout in [(int(v[0]), int(v[1])) for v in a.variants.split(",")]:       # points on device?, labels on device?
    ctxs = [[P.Context(0) for _ in range(a.batch)] for _ in range(a.groups)]
    if pin: pts = [[torch.from_numpy(src[i % 4]).to(dev) for i in range(a.batch)] for _ in range(a.groups)]
    else: pts = [[torch.from_numpy(src[i % 4]).pin_memory() for i in range(a.batch)] for _ in range(a.groups)]
    if lout: lab = [[torch.empty(npts, dtype=torch.int32, device=dev) for _ in range(a.batch)] for _ in range(a.groups)]
    else: lab = [[torch.empty(npts, dtype=torch.int32).pin_memory() for _ in range(a.batch)] for _ in range(a.groups)]
    torch.cuda.synchronize()
    stage = {}
    def worker(g, rounds):
        time.sleep(a.stagger_ms * g / 1e3)
        k = a.batch
        handles = (vp * k)(*[c.handle for c in ctxs[g]]); results = (P.Result * k)()
        pp = (vp * k)(*[vp(t.data_ptr()) for t in pts[g]]); lp = (vp * k)(*[vp(t.data_ptr()) for t in lab[g]])
        cnt = (ctypes.c_size_t * k)(*[npts] * k)
        for _ in range(rounds):
            rc = lib.f3ds_segment_batch(handles, k, pp, cnt, pin, ctypes.byref(prm), lp, lout, results)
            assert rc == 0, rc
        stage[g] = [results[0].ms_stage[j] for j in range(7)]
    for rounds in (1, a.rounds):
        th = [threading.Thread(target=worker, args=(g, rounds)) for g in range(a.groups)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
    print("points %s, labels %s: %7.1f frames/s" % ("on device" if pin else "host pinned", "on device" if lout else "host pinned", a.groups * a.batch * a.rounds / dt), " stage ms of the last batch of group 0:", " ".join("%.0f" % x for x in stage[0]), flush=True)
    for grp in ctxs:
        for c in grp: c.close()
    del pts, lab
