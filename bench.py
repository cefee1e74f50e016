#!/usr/bin/env python3
"""bench.py -- Mpoints/s segmented end-to-end on synthetic 1M-point RGB-D frames (BASELINE.json).

A *step* is one pass of the whole hot path (voxelise -> normals -> seeds -> sweeps -> adjacency ->
merge -> per-point labels) over one 1,000,000-point XYZRGBA frame that is already resident in
HBM when the timed region starts (config 2 of BASELINE.md: 1000x1000 pinhole frame, 3 % NaN,
flags ``-v 0.008 -s 0.08 --AL --CVX -t 0.2``).  Frames are independent, and a single frame holds
two inherently sequential pieces (16 Gauss-Seidel sweeps, ~2800 dependent merges), so the steps
are issued ``--batch`` frames at a time through ``f3ds_segment_batch`` -- every kernel of the path
is then ONE dispatch for the whole batch (grid.y = frame; the merge loops run one workgroup per
frame) -- and ``--groups`` such calls are in flight from host threads so that one batch's wide
kernels overlap another batch's merge dispatch.  Every step still runs the complete path on its
own frame; the JSON also reports the latency of a single frame with nothing else in flight.

N > 1 (launched by ``python -m torch.distributed.run``): one process per GPU, each rank segments
its own frames (weak scaling, no data-path collective); the label output is one RCCL gather of
the uint32 label buffers to rank 0 per frame.

The JSON line also carries
  roofline     -- dominant kernel by device time (HIP events recorded on the batch's stream inside
                  libf3ds around each stage): achieved = 20 B/point x points per launch / mean launch
                  duration, against the 8 TB/s HBM3E peak;
  cpu_baseline -- the CPU oracle (single thread, kind "port": the reference needs PCL/OpenCV and
                  cannot be built here) on one frame of the same workload, rank 0 at N=1 only.
"""
import argparse
import ctypes
import importlib
import json
import os
import queue
import sys
import threading
import time

# HIP maps streams onto 4 hardware queues by default; a frame's merge kernel is one long
# single-workgroup launch, so with more frames in flight than queues a second frame's short kernels
# would wait behind it.  Must be set before the HIP runtime starts (22.8 -> 44 Mpoints/s in round 1).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

STAGES = ["voxelise", "neighbours+normals", "seeds", "sweeps", "summaries+adjacency+weights", "merge", "labels"]
# the kernel that dominates each stage (rocprofv3 --kernel-trace names in profiles/)
STAGE_KERNEL = {"voxelise": "k_batched<d_radix_scatter>", "neighbours+normals": "k_batched<d_normals>", "seeds": "k_batched<d_seed_nn>", "sweeps": "k_batched<d_sweep_R>",
                "summaries+adjacency+weights": "k_batched<d_sv_fill>", "merge": "k_batched<d_merge_lds_t<true>>", "labels": "k_batched<d_point_labels>"}
ALG_BYTES_PER_POINT = 20          # 16 B read of {x,y,z,rgba} + 4 B label write
HBM_PEAK_GBS = 8000.0             # MI355X HBM3E (guides/MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3072)
    ap.add_argument("--warmup", type=int, default=768)
    ap.add_argument("--batch", type=int, default=192, help="frames per f3ds_segment_batch call")
    ap.add_argument("--groups", type=int, default=4, help="batch calls in flight per GPU (libf3ds runs up to four on distinct hardware queues)")
    ap.add_argument("--frames", type=int, default=4, help="distinct synthetic frames to cycle through")
    ap.add_argument("--width", type=int, default=1000)
    ap.add_argument("--height", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # development: F3DS_BENCH_FORCE_DIST=1 takes the RCCL path (process group, label gather, barrier) with one rank too
    dist_on = world > 1 or bool(os.environ.get("F3DS_BENCH_FORCE_DIST"))
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libf3ds has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    P = importlib.import_module("fast-3d-pointcloud-segmentation_amd")
    B = importlib.import_module("fast-3d-pointcloud-segmentation_amd.batch")
    prm = P.launch_params(voxel_res=0.008, seed_res=0.08)          # -v 0.008 -s 0.08 --AL --CVX -t 0.2
    if os.environ.get("F3DS_BENCH_THRESHOLD"):                    # development only: what-if runs (the JSON line then names the threshold)
        prm.threshold = float(os.environ["F3DS_BENCH_THRESHOLD"])
    npts = args.width * args.height

    # synthetic frames -> HBM (torch owns the device buffers; libf3ds gets raw pointers)
    frames_host = [P.synth_frame(0, 1000 + rank * 64 + i, args.width, args.height, 30) for i in range(args.frames)]
    frames_dev = [torch.from_numpy(f).to(dev) for f in frames_host]
    # frames in flight = groups x batch: every group segments `batch` frames per f3ds_segment_batch call
    # (wide stages on one stream per frame, all merge loops of the batch in ONE dispatch); `groups` such
    # calls run concurrently from host threads so one batch's wide stages overlap another's merge loops.
    nbatch, ngroups = max(1, args.batch), max(1, args.groups)
    if args.steps < nbatch * ngroups:       # few steps: spread them over the groups instead of leaving groups idle
        nbatch = max(1, -(-args.steps // ngroups))
    nstreams = nbatch * ngroups
    ctxs = [[P.Context(local_rank) for _ in range(nbatch)] for _ in range(ngroups)]
    # one contiguous label block per group: the batch's label output is ONE RCCL gather (nbatch x 4 MB per rank)
    label_blocks = [torch.empty((nbatch, npts), dtype=torch.int32, device=dev) for _ in range(ngroups)]
    label_bufs = [[label_blocks[g][i] for i in range(nbatch)] for g in range(ngroups)]
    gather_list = [torch.empty((nbatch, npts), dtype=torch.int32, device=dev) for _ in range(world)] if (dist_on and rank == 0) else None
    torch.cuda.synchronize()

    stage_ms = [0.0] * 7
    batches_done, frames_done = [0], [0]
    stage_lock = threading.Lock()
    errors = []

    def run_steps(count, record):
        q = queue.Queue()
        for s0 in range(0, count, nbatch):
            q.put(list(range(s0, min(count, s0 + nbatch))))

        def worker(g):
            torch.cuda.set_device(local_rank)
            while True:
                try:
                    steps = q.get_nowait()
                except queue.Empty:
                    return
                try:
                    k = len(steps)
                    P.segment_batch(ctxs[g][:k], [frames_dev[s % len(frames_dev)].data_ptr() for s in steps], prm,
                                    labels_out=[label_bufs[g][i].data_ptr() for i in range(k)], n=[npts] * k, on_device=True)
                    if dist_on:     # label output of this batch: one RCCL gather of the whole label block to rank 0
                        with stage_lock:
                            B.gather_label_block(label_blocks[g], dist, gather_list, dst=0)
                            torch.cuda.current_stream().synchronize()      # the block is reused by this group's next batch (on libf3ds' own stream)
                    if record:      # ms_stage is the device time of each stage of the whole batch (HIP events on the batch stream)
                        with stage_lock:
                            for j in range(7):
                                stage_ms[j] += ctxs[g][0].result.ms_stage[j]
                            batches_done[0] += 1
                            frames_done[0] += k
                except Exception as e:   # noqa
                    errors.append(e)
                    return

        threads = [threading.Thread(target=worker, args=(g,)) for g in range(ngroups)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, not warm-up: every context grows its device scratch on first use (hipMalloc), so each is used once before
    # the W warm-up steps, however small W is; the timed region never allocates
    run_steps(nbatch * ngroups, False)
    barrier()
    run_steps(args.warmup, False)
    barrier()
    if os.environ.get("F3DS_BENCH_MEMINFO"):                      # development: HBM in use after warm-up (contexts are grow-only)
        free, total = torch.cuda.mem_get_info(dev)
        print("rank %d: %.1f GB of %.1f GB HBM in use" % (rank, (total - free) / 2**30, total / 2**30), file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    run_steps(args.steps, True)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # single-frame latency (one stream, nothing else in flight) for the record
    barrier()
    tl = time.perf_counter()
    for s in range(3):
        ctxs[0][0].segment(frames_dev[s % len(frames_dev)].data_ptr(), prm, labels_out=label_bufs[0][0].data_ptr(), n=npts, on_device=True)
    latency_ms = (time.perf_counter() - tl) / 3 * 1e3
    res = ctxs[0][0].result

    if rank == 0:
        value = world * args.steps * npts / elapsed / 1e6
        mean_stage = [m / max(1, batches_done[0]) for m in stage_ms]       # per batched launch sequence
        frames_per_launch = frames_done[0] / max(1, batches_done[0])
        # The dominant KERNEL is the merge loop: 42 % of all device time in profiles/r1_kernel_stats.csv, one launch
        # per batch, so its launch duration is exactly the 'merge' stage measured live below.  (The 'sweeps' stage
        # can be as long, but it is 64 launches of four different kernels.)
        dom = STAGES.index("merge")
        dom_ms = mean_stage[dom]
        # one launch of the dominant kernel processes `frames_per_launch` frames (grid.y = frame)
        achieved = ALG_BYTES_PER_POINT * npts * frames_per_launch / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = None
        try:        # HBM bytes per launch of the dominant kernel from the committed PMC passes of this same command
            pm = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_hbm_traffic.json")))
            kname = STAGE_KERNEL[STAGES[dom]].split("<")[1].rstrip(">")      # d_merge_lds_t, d_sweep_R, ...
            if kname in pm["kernels"]:
                traffic = int(pm["kernels"][kname]["hbm_bytes_per_frame"] * frames_per_launch)
        except Exception:
            traffic = None
        roofline = {"bound": "hbm", "kernel": STAGE_KERNEL[STAGES[dom]], "stage": STAGES[dom], "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                    "launch_ms": round(dom_ms, 4), "frames_per_launch": frames_per_launch,
                    "algorithmic_bytes_per_launch": int(ALG_BYTES_PER_POINT * npts * frames_per_launch),
                    "whole_path_achieved_GBps": round(ALG_BYTES_PER_POINT * npts * frames_per_launch / (sum(mean_stage) * 1e-3) / 1e9, 3) if sum(mean_stage) > 0 else None,
                    "stage_ms": {STAGES[i]: round(mean_stage[i], 4) for i in range(7)}}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from conftest import CpuChecker
            ora = CpuChecker(os.path.join(ROOT, "oracle", "libf3ds_oracle.so"), "f3ds_oracle")
            tc = time.perf_counter()
            rc, olab, ores, oh = ora.segment(frames_host[0], prm)
            cpu_s = time.perf_counter() - tc
            assert rc == 0
            cpu = {"value": round(npts / cpu_s / 1e6, 4), "unit": "Mpoints/s", "cores": 1, "kind": "port",
                   "sample": "1 frame (%d points) of the same workload through oracle/libf3ds_oracle.so, %.2f s" % (npts, cpu_s),
                   "host_cpus": os.cpu_count()}
            oh.close()
        line = {"metric": "Mpoints/sec segmented end-to-end, 1M-pt RGB-D frames", "value": round(value, 3), "unit": "Mpoints/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "single synthetic %dx%d (%d-point) XYZRGBA frame per step, -v 0.008 -s 0.08 --AL --CVX -t %g" % (args.width, args.height, npts, prm.threshold),
                           "frames_in_flight_per_gpu": nstreams, "setup": "one untimed pass over all contexts (scratch allocation) before the warm-up steps", "batch": nbatch, "concurrent_batches": ngroups, "distinct_frames": args.frames, "parallelism": "frames sharded one per GPU" if world > 1 else "1 GPU",
                           "label_gather": "one RCCL gather of the batch's label block (batch x 4 MB per rank) to rank 0 per batch" if dist_on else "none",
                           "V": res.n_voxels, "S": res.n_supervoxels, "E": res.n_edges, "merges": res.n_merges, "regions": res.n_regions},
                "single_stream_latency_ms": round(latency_ms, 3), "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(line), flush=True)
    for grp in ctxs:
        for c in grp:
            c.close()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
