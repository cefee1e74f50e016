#!/usr/bin/env python3
"""bench.py -- Mpoints/s segmented end-to-end on synthetic 1M-point RGB-D frames (BASELINE.json).

A *step* is one pass of the whole hot path (voxelise -> normals -> seeds -> sweeps -> adjacency ->
merge -> per-point labels) over ONE BATCH OF 64 DISTINCT 1,000,000-point XYZRGBA frames per GPU --
BASELINE.json config 5's batch (64 frames, seeds 1000..1063 on rank 0; rank r takes seeds
1000+64r..) on every GPU, each frame the 1000x1000 pinhole frame of config 2 with 3 % NaN,
flags ``-v 0.008 -s 0.08 --AL --CVX -t 0.2`` -- resident in HBM when the timed region starts.
Frames are independent, and a single frame holds two inherently sequential pieces (16 Gauss-Seidel
sweeps, ~2400 dependent merges), so the frames of the K timed steps are issued through
``f3ds_segment_batch`` in calls of up to ``--batch`` frames (every kernel of the path is then ONE
dispatch for the whole call, grid.y = frame) and ``--groups`` such calls are in flight from host
threads so that one call's wide kernels overlap another call's merge dispatch.  Every frame of
every step runs the complete path; nothing is cached between steps.

``--gpus N`` with N > 1 and no torchrun environment: this process starts the N ranks itself (a child
``python -m torch.distributed.run``; nothing here touches the GPU before that).  One process per GPU,
every rank runs its own 64 frames per step (weak scaling, no data-path collective); the label
output of a step is ONE RCCL gather of the step's [64, 1M] uint32 label block to rank 0, issued in
step order from a dedicated thread while later steps run.
``--strong``: BASELINE config 5 as literally written instead -- ONE batch of 64 frames (seeds 1000..1063) per step for the whole
job, frame i on rank i mod N (64 / N frames per rank and step), the step's label blocks gathered to rank 0 in ONE RCCL gather;
`scaling` is then "strong" and `value` still the whole job's points per second.

The JSON line also carries
  value              -- whole-job Mpoints/s with the frames resident in HBM when the timed region starts and the
                        labels left in HBM (the bench contract of this build: a PCIe-inclusive rate is never `value`);
                        `value_hbm_resident` repeats it under an explicit name;
  value_host_io      -- the metric as SURVEY.md 8d words it (host buffer in -> per-point labels in a host buffer):
                        the same loop with the frames in pinned host memory and the labels delivered to pinned host
                        memory, PCIe both ways inside the timed region; `value_survey_8d` repeats its number at top level;
  config5_batch_latency_ms -- BASELINE config 5 as literally stated, from an idle GPU: wall time of ONE batch of 64 frames
                        as one call on one GPU (the N=1 shape) and of ONE call of 8 frames (what each GPU runs at N=8);
  roofline           -- the dominant kernel is the merge loop (k_batched<d_merge_*>: first by total kernel time in the
                        rocprofv3 trace of this same command, the newest profiles/r*_kernel_stats.csv); its launch duration is
                        measured live by a pair of HIP events around that one dispatch on the call's stream inside
                        libf3ds (f3ds_result.ms_stage[5]) and averaged over the timed region's calls.  achieved =
                        20 B/point x points per launch / mean launch duration, against the 8 TB/s HBM3E peak;
                        `stages` = the same figure for every stage of a call (sequences of many launches);
                        `whole_path` = the timed region as a whole, with the PMC-measured HBM traffic of all kernels;
  cpu_baseline       -- the CPU oracle (kind "port": the reference needs PCL/OpenCV and cannot be built here) on the
                        same workload, rank 0 at N=1 only: one frame on one core (the reference is single-threaded)
                        through the libm-linked build of the oracle -- what a PCL/OpenCV build calls; its labels are
                        first checked against the shared-math build's, whose time is reported beside it --, and
                        `frames_parallel`: the 64-frame batch, one frame per core, over the host's cores (count stated).
A label hash that differs from the oracle's committed hashes voids the run: `value` is null and the exit code 1; so
does any what-if environment that changes results or the measured configuration (F3DS_BENCH_THRESHOLD, a +whatif library).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
FRAMES_PER_STEP = 64              # BASELINE.json config 5: a batch of 64 synthetic 1M-point frames
STAGES = ["voxelise", "neighbours+normals", "seeds", "sweeps", "summaries+adjacency+weights", "merge", "labels"]
ALG_BYTES_PER_POINT = 20          # 16 B read of {x,y,z,rgba} + 4 B label write
HBM_PEAK_GBS = 8000.0             # MI355X HBM3E (guides/MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48, help="timed steps; a step = 64 distinct 1M-point frames per GPU")
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--batch", type=int, default=192, help="most frames per f3ds_segment_batch call")
    ap.add_argument("--groups", type=int, default=6, help="batch calls in flight per GPU (libf3ds lends each a stream of its own; measured 4: 2 080, 6: 2 200, 7: 1 990, 8: 1 820 Mpoints/s at --steps 20)")
    ap.add_argument("--width", type=int, default=1000)
    ap.add_argument("--height", type=int, default=1000)
    ap.add_argument("--host-io-steps", type=int, default=-1, help="steps of the pinned-host-in / host-out pass (default min(steps, 12); 0 = skip)")
    ap.add_argument("--host-io-groups", type=int, default=0, help="batch calls in flight in the host-in / host-out pass (0 = as --groups): a call is longer there by its PCIe copies")
    ap.add_argument("--strong", action="store_true", help="strong scaling: one 64-frame batch per step for the whole job, frame i on rank i mod N (BASELINE config 5 as written); default: weak, 64 frames per step on every rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--skip-latency", action="store_true", help="no lone-frame / one-batch latency runs after the timed region (profile runs: only the timed loop's launches in the trace)")
    return ap.parse_args()


def cpu_baseline_record(npts, cpu_s, sm_s, main_s, libm_same, host_cpus):
    """The `cpu_baseline` object of the JSON line from the oracle's timings (one frame, one core).  The baseline is timed on the libm-linked
    build; a label vector that differs from the shared-math build's (the parity checker) voids it: value null, reason stated."""
    cpu = {"value": round(npts / cpu_s / 1e6, 4), "unit": "Mpoints/s", "cores": 1, "kind": "port",
           "sample": "1 frame (%d points) of the same workload through oracle/libf3ds_oracle_libm.so (libm transcendentals), %.2f s" % (npts, cpu_s),
           "labels_equal_shared_math_build": bool(libm_same),
           "shared_math_build": {"value": round(npts / sm_s / 1e6, 4), "seconds": round(sm_s, 2), "what": "the same frame through oracle/libf3ds_oracle.so (the bit-parity checker: transcendentals of csrc/f3ds_math.h)"},
           "as_main_runs_it": {"value": round(npts / main_s / 1e6, 4), "seconds": round(main_s, 2),
                               "what": "label path + refineSupervoxels(3) + a second VCCS extract (the truth cloud), as main() does per file"},
           "note": "the port is faster than the reference would be: hash-set contains() instead of the O(E) scan, cached mean_color",
           "host_cpus": host_cpus}
    if not libm_same:
        cpu["value"] = None
        cpu["as_main_runs_it"]["value"] = None
        cpu["invalid"] = "the libm-linked oracle's labels differ from the shared-math oracle's on this frame"
    return cpu


def strong_estimate(batch_latency, npts):
    """What ONE 64-frame batch (config 5 as written) would take on N GPUs, from this GPU's own measurements -- an estimate, not a measurement: no
    multi-GPU node has run it.  At N = 8 every GPU runs one call of 8 frames (`one_call_of_8_frames`, measured here from an idle GPU); the label
    gather (8 x 4 MB per peer over its own xGMI link, ~0.2 ms) is not in it."""
    if not batch_latency:
        return None
    b64, b8 = batch_latency["one_batch_of_64_frames_one_gpu"], batch_latency["one_call_of_8_frames"]
    return {"n1_ms_measured": b64, "n8_ms_estimated": b8, "speedup_8_gpus_estimated": round(b64 / b8, 2) if b8 > 0 else None,
            "mpoints_per_s_n8_estimated": round(FRAMES_PER_STEP * npts / b8 / 1e3, 1) if b8 > 0 else None,
            "what": "lone-batch latency of config 5 as written; UNMEASURED on hardware for N > 1 (1-GPU pool): a lone call's merge loop (~30 ms) does not shrink with fewer frames per GPU"}


def spawn_ranks(args):
    """--gpus N without a torchrun environment: start the N ranks as a CHILD job and relay its exit code.  This parent
    never imports torch or touches HIP (a process that has initialised the GPU must not be replaced or re-launched)."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    # HIP maps streams onto 4 hardware queues by default; libf3ds keeps one stream per queue for its batch calls.
    # Must be set before the HIP runtime starts.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    from concurrent.futures import ThreadPoolExecutor
    import threading

    import numpy as np      # noqa
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: running %d rank(s)" % (args.gpus, world, world), file=sys.stderr, flush=True)
    # development: F3DS_BENCH_FORCE_DIST=1 takes the RCCL path (process group, per-step label gather, barrier) with one rank too
    dist_on = world > 1 or bool(os.environ.get("F3DS_BENCH_FORCE_DIST"))
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "RANK" not in os.environ:      # single forced rank without torchrun
            import socket
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(port))
            os.environ["RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libf3ds has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    P = importlib.import_module("fast-3d-pointcloud-segmentation_amd")
    B = importlib.import_module("fast-3d-pointcloud-segmentation_amd.batch")
    P.check_library_is_current()       # a stale prebuilt libf3ds.so (built from other sources than those beside it) is not measured
    lib_text = P.library_stamp()[1]
    prm = P.launch_params(voxel_res=0.008, seed_res=0.08)          # -v 0.008 -s 0.08 --AL --CVX -t 0.2
    if os.environ.get("F3DS_BENCH_THRESHOLD"):                    # development only: what-if runs (the JSON line then names the threshold)
        prm.threshold = float(os.environ["F3DS_BENCH_THRESHOLD"])
    npts = args.width * args.height
    if args.strong:      # config 5 as written: the step's 64 frames sharded i mod N
        if FRAMES_PER_STEP % world:
            raise SystemExit("bench.py --strong: %d ranks do not divide the batch of %d frames" % (world, FRAMES_PER_STEP))
        mine = B.frames_of_rank(FRAMES_PER_STEP, rank, world)
        FPS = len(mine)                                   # frames of a step THIS rank runs
        seeds = [1000 + i for i in mine]
    else:                # weak: every rank its own 64 distinct frames per step
        FPS = FRAMES_PER_STEP
        seeds = [1000 + rank * FPS + i for i in range(FPS)]

    # the rank's distinct synthetic frames -> HBM (torch owns the device buffers; libf3ds gets raw pointers)
    with ThreadPoolExecutor(8) as ex:
        frames_host = list(ex.map(lambda s: P.synth_frame(0, s, args.width, args.height, 30), seeds))
    frames_dev = [torch.from_numpy(f).to(dev) for f in frames_host]
    nbatch, ngroups = max(1, args.batch), max(1, args.groups)
    hgroups = args.host_io_groups if args.host_io_groups > 0 else ngroups
    ctxs = [[P.Context(local_rank) for _ in range(nbatch)] for _ in range(max(ngroups, hgroups))]
    # label output: a ring of per-step blocks [64, npts]; a step's block is ONE RCCL gather (64 x 4 MB per rank)
    n_blocks = -(-(nbatch * max(ngroups, hgroups)) // FPS) + 2
    label_blocks = [torch.empty((FPS, npts), dtype=torch.int32, device=dev) for _ in range(n_blocks)]
    gather_list = [torch.empty((FPS, npts), dtype=torch.int32, device=dev) for _ in range(world)] if (dist_on and rank == 0) else None
    gather_stream = torch.cuda.Stream(device=dev) if dist_on else None
    torch.cuda.synchronize()

    stage_ms = [0.0] * 8
    calls_done, frames_done = [0], [0]
    stat_lock = threading.Lock()
    mode = {"record": False, "host": False}
    host_frames, host_labels = [], []

    def run_batch(g, f0, f1):
        torch.cuda.set_device(local_rank)
        k = f1 - f0
        if mode["host"]:
            pts = [host_frames[f % FPS].data_ptr() for f in range(f0, f1)]
            out = [host_labels[(f // FPS) % n_blocks][f % FPS].data_ptr() for f in range(f0, f1)]
            P.segment_batch(ctxs[g][:k], pts, prm, labels_out=out, n=[npts] * k, raw_host=True)
        else:
            pts = [frames_dev[f % FPS].data_ptr() for f in range(f0, f1)]
            out = [label_blocks[(f // FPS) % n_blocks][f % FPS].data_ptr() for f in range(f0, f1)]
            P.segment_batch(ctxs[g][:k], pts, prm, labels_out=out, n=[npts] * k, on_device=True)
        if mode["record"]:      # ms_stage is the device time of each stage of the whole call (HIP events on the call's stream)
            with stat_lock:
                for j in range(8):
                    stage_ms[j] += ctxs[g][0].result.ms_stage[j]
                calls_done[0] += 1
                frames_done[0] += k

    def gather_step(s):         # label output of step s: one RCCL gather of its label block to rank 0 (in step order on every rank)
        torch.cuda.set_device(local_rank)
        with torch.cuda.stream(gather_stream):
            B.gather_label_block(label_blocks[s % n_blocks], dist, gather_list, dst=0)
        gather_stream.synchronize()     # the ring slot is written again by a later step (on libf3ds' own streams)

    ramp = os.environ.get("F3DS_BENCH_RAMP", "1") != "0"       # development: 0 = equal calls (see plan_batches)
    pipe = B.StepPipeline(FPS, nbatch, ngroups, n_blocks, run_batch, gather_step if dist_on else None, ramp=ramp)

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, not warm-up: every context grows its device scratch on first use (hipMalloc), so each is used once before
    # the W warm-up steps, however small W is; the timed region never allocates
    def use_every_context():
        ts = [threading.Thread(target=run_batch, args=(g, g * nbatch, (g + 1) * nbatch)) for g in range(len(ctxs))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()

    use_every_context()
    use_every_context()       # second use: every context's scratch goes to the device-wide high-water marks of the first pass
    barrier()
    pipe.run(args.warmup)
    barrier()
    if os.environ.get("F3DS_BENCH_MEMINFO"):                      # development: HBM in use after warm-up (contexts are grow-only)
        free, total = torch.cuda.mem_get_info(dev)
        print("rank %d: %.1f GB of %.1f GB HBM in use" % (rank, (total - free) / 2**30, total / 2**30), file=sys.stderr, flush=True)
    mode["record"] = True
    t0 = time.perf_counter()
    plan = pipe.run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    mode["record"] = False
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the same loop, frames from pinned host memory, labels into pinned host memory (PCIe inside the timed region)
    hio_steps = min(args.steps, 12) if args.host_io_steps < 0 else args.host_io_steps
    host_io = None
    if hio_steps > 0:
        host_frames.extend(torch.from_numpy(f).pin_memory() for f in frames_host)
        host_labels.extend(torch.empty((FPS, npts), dtype=torch.int32).pin_memory() for _ in range(n_blocks))
        mode["host"] = True
        hpipe = B.StepPipeline(FPS, nbatch, hgroups, n_blocks, run_batch, None, ramp=ramp)
        use_every_context()                                 # contexts allocate their upload buffers once
        barrier()
        th = time.perf_counter()
        hpipe.run(hio_steps)
        barrier()
        el_h = time.perf_counter() - th
        if dist_on:
            t = torch.tensor([el_h], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el_h = float(t.item())
        host_io = {"value": round(world * hio_steps * FPS * npts / el_h / 1e6, 3), "unit": "Mpoints/s", "steps": hio_steps,
                   "concurrent_calls": hgroups,
                   "what": "same loop, frames in pinned host memory, labels delivered to pinned host memory (no RCCL gather in this pass)"}
        mode["host"] = False
        # What bounds this number: the host link.  Measured here, in the same run (pinned memory, 256 MB per copy): one way alone, and both ways at once on two
        # streams.  The library queues every call's uploads and downloads on ONE copy stream per device (include/f3ds.h): 20 MB per frame one copy after the other is
        # the bound that schedule is to be read against (`..._if_copies_serialise`); `..._if_both_ways_overlap` is what a schedule that kept both directions busy
        # could reach at the duplex rate measured beside it (on some boxes of the pool the link carries LESS in total with both directions active than one way alone --
        # 2 x 16 against 55 GB/s, DESIGN.md 8 --, on others nearly twice as much: the line says which kind this box is).
        try:
            n = 256 << 20
            hbuf = torch.empty(n, dtype=torch.uint8).pin_memory(); hbuf2 = torch.empty(n, dtype=torch.uint8).pin_memory()
            dbuf = torch.empty(n, dtype=torch.uint8, device=dev); dbuf2 = torch.empty(n, dtype=torch.uint8, device=dev)

            def rate(fn, reps=4):
                fn(); torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize(dev)
                return reps * n / (time.perf_counter() - t0) / 1e9
            h2d = rate(lambda: dbuf.copy_(hbuf, non_blocking=True)); d2h = rate(lambda: hbuf.copy_(dbuf, non_blocking=True))
            s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

            def both():
                with torch.cuda.stream(s1):
                    dbuf.copy_(hbuf, non_blocking=True)
                with torch.cuda.stream(s2):
                    hbuf2.copy_(dbuf2, non_blocking=True)
            duplex = rate(both)
            bytes_in, bytes_out = 16.0 * npts, 4.0 * npts
            bound = world * npts / (bytes_in / (h2d * 1e9) + bytes_out / (d2h * 1e9)) / 1e6
            bound2 = world * npts / (max(bytes_in, bytes_out) / (duplex * 1e9)) / 1e6
            host_io["link_bound"] = {"h2d_GBps": round(h2d, 1), "d2h_GBps": round(d2h, 1), "both_ways_at_once_GBps_each": round(duplex, 1),
                                     "Mpoints_per_s_if_copies_serialise": round(bound, 1), "fraction_of_bound": round(host_io["value"] / bound, 3),
                                     "Mpoints_per_s_if_both_ways_overlap": round(bound2, 1),
                                     "what": "16 B in + 4 B out per point over the link rates measured in this run: one copy after the other at the one-way rates (the library's "
                                             "schedule: one copy stream per device), and uploads beside downloads at the rate each direction reaches when both are active"}
            del hbuf, hbuf2, dbuf, dbuf2
        except Exception as ex:      # noqa (a measurement beside the line: never fails the run)
            host_io["link_bound"] = {"error": repr(ex)}

    # the labels the timed region produced, against the oracle's committed hashes (tests/golden/oracle_golden_big.json, made in the
    # build container by tools/make_golden_big.py): the last step's block holds the frames of seeds 1000 + 64 rank + i.  Every frame of
    # that block is checked on every rank; a mismatch voids the run (value null, exit code 1).
    parity = None
    try:
        import hashlib
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_golden_big.json")))
        blk = label_blocks[(args.steps - 1) % n_blocks].cpu().numpy()
        checked, bad = 0, []
        if npts == 1000 * 1000 and not os.environ.get("F3DS_BENCH_THRESHOLD"):      # (another threshold gives other labels: such a run is void anyway, below)
            with ThreadPoolExecutor(16) as ex:
                keys = ["config5_seed%d" % sd for sd in seeds]
                have = [i for i in range(FPS) if keys[i] in gold]
                digests = list(ex.map(lambda i: hashlib.sha256(blk[i].tobytes()).hexdigest(), have))
            for i, dg in zip(have, digests):
                checked += 1
                if dg != gold[keys[i]]["labels_sha256"]:
                    bad.append(seeds[i])
        if dist_on:
            t = torch.tensor([checked, len(bad)], dtype=torch.int64, device=dev)
            dist.all_reduce(t)
            checked_all, bad_all = int(t[0].item()), int(t[1].item())
        else:
            checked_all, bad_all = checked, len(bad)
        if checked_all:
            parity = {"frames_checked": checked_all, "mismatches": bad, "mismatches_all_ranks": bad_all,
                      "against": "SHA-256 of the oracle's per-point labels (tests/golden/oracle_golden_big.json), every frame of the last timed step's label block on every rank"}
    except Exception as e:      # noqa
        parity = {"error": repr(e)}

    merge_layouts = {}      # which merge kernel every call group's last batch call ran (before the lone frames below overwrite it)
    for grp in ctxs[:ngroups]:
        try:
            lay = grp[0].merge_layout()
            merge_layouts[lay] = merge_layouts.get(lay, 0) + 1
        except Exception:
            pass
    # single-frame latency (one stream, nothing else in flight) for the record
    barrier()
    tl = time.perf_counter()
    for s in range(0 if args.skip_latency else 3):
        ctxs[0][0].segment(frames_dev[s].data_ptr(), prm, labels_out=label_blocks[0][0].data_ptr(), n=npts, on_device=True)
    latency_ms = (time.perf_counter() - tl) / 3 * 1e3 if not args.skip_latency else 0.0
    res = ctxs[0][0].result
    # BASELINE config 5 as literally stated ("a batch of 64 frames sharded over 8 GPUs"), from an idle GPU: ONE batch of 64 frames as one
    # call on this GPU (N = 1), and ONE call of 8 frames (each GPU's share at N = 8); frames and labels in HBM; median of 3 after one untimed call
    def one_call_ms(k):
        times = []
        for rep_ in range(4):
            barrier()
            t1 = time.perf_counter()
            P.segment_batch(ctxs[0][:k], [frames_dev[f].data_ptr() for f in range(k)], prm, labels_out=[label_blocks[0][f].data_ptr() for f in range(k)], n=[npts] * k, on_device=True)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t1) * 1e3)
        return sorted(times[1:])[1]
    batch_latency = None
    if nbatch >= FRAMES_PER_STEP and FPS == FRAMES_PER_STEP and not args.skip_latency:      # (a --strong rank at N > 1 holds fewer than 64 frames)
        b64, b8 = one_call_ms(FPS), one_call_ms(8)
        batch_latency = {"one_batch_of_64_frames_one_gpu": round(b64, 3), "one_call_of_8_frames": round(b8, 3),
                         "mpoints_per_s_64": round(FPS * npts / b64 / 1e3, 1), "mpoints_per_s_8": round(8 * npts / b8 / 1e3, 1),
                         "what": "wall time of ONE f3ds_segment_batch call from an idle GPU, frames and labels resident in HBM, median of 3: config 5's batch on one GPU, and the 8 frames each GPU runs when the 64 are sharded over 8 GPUs (the RCCL gather of 8 x 4 MB per peer is not in it)"}

    if rank == 0:
        total_frames = args.steps * FPS
        value = world * total_frames * npts / elapsed / 1e6
        mean_stage = [m / max(1, calls_done[0]) for m in stage_ms]       # per batched launch sequence
        frames_per_launch = frames_done[0] / max(1, calls_done[0])
        # The dominant KERNEL: the merge loop is ONE launch per call and timed on its own (f3ds_result.ms_stage[5]: a HIP event pair on the call's stream around
        # that dispatch).  The other stages are sequences of many launches of several kernels (per-kernel split: the newest profiles/r*_kernel_stats*.csv) and
        # are reported as stages below; ms_stage[7], the event pair around the voxel-normal dispatch, also counts the time that dispatch waits at the head of
        # its queue for a compute unit with room for its first workgroup, so it is listed among the stages and never used to rank kernels.
        # (which merge kernel: the library takes the 4-wave layout for a call that shares the device with other batch calls, the 8-wave one otherwise; asked of every
        # call group's context after the timed region (above) -- what its LAST call ran -- and the most frequent answer names the kernel)
        layouts = merge_layouts
        lay = max(layouts, key=layouts.get) if layouts else (8, 2)
        merge_name = "k_batched<d_merge_il_t<%d,%d>>" % lay if lay[0] else "k_batched<d_merge>"
        KERNELS = {5: (merge_name, "merge", "d_merge_il_t"), 7: ("k_batched<d_normals_t<%d>>" % (256 if nbatch >= 16 else 384), "neighbours+normals", "d_normals_t")}      # (256 threads per tile in calls of >= 16 frames)
        # the merge loop is the first kernel by total time in the kernel trace of this command (the newest profiles/r*_kernel_stats.csv), and the one kernel whose
        # event pair brackets exactly one dispatch
        dom = 5
        dom_ms = mean_stage[dom]
        alg_launch = ALG_BYTES_PER_POINT * npts * frames_per_launch
        achieved = alg_launch / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = path_traffic = path_traffic_min = None
        pmc_file = None
        for cand in ("r5_pmc_hbm_traffic.json", "r4_pmc_hbm_traffic.json", "r3_pmc_hbm_traffic.json", "r2_pmc_hbm_traffic.json"):      # HBM bytes from the committed PMC passes of this same command (profiles/, tools/pmc_summary.py)
            if os.path.exists(os.path.join(ROOT, "profiles", cand)):
                pmc_file = cand
                break
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
            pk = pm["kernels"]
            traffic = int((pk.get(KERNELS[dom][2]) or pk.get("d_merge_cw_t") or pk[KERNELS[dom][2].replace("_t", "")])["hbm_bytes_per_frame"] * frames_per_launch)
            path_traffic = int(pm["whole_path_hbm_bytes_per_frame"])          # upper bound (every read request taken as 128 bytes)
            path_traffic_min = int(pm["whole_path_hbm_bytes_per_frame_min"])  # lower bound (64-byte requests in the kernels that gather)
        except Exception:
            pass
        whole = {"achieved": round(ALG_BYTES_PER_POINT * value * 1e6 / world / 1e9, 3), "unit": "GB/s", "frac": round(ALG_BYTES_PER_POINT * value * 1e6 / world / 1e9 / HBM_PEAK_GBS, 6),
                 "algorithmic_bytes_per_frame": ALG_BYTES_PER_POINT * npts, "traffic_per_frame": path_traffic, "traffic_per_frame_min": path_traffic_min,
                 "wasted_ratio": round(path_traffic / (ALG_BYTES_PER_POINT * npts), 2) if path_traffic else None,
                 "traffic_source": "profiles/%s: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over every launch of the path (one call at a time), units calibrated in profiles/r2_pmc_calibration.json" % pmc_file}

        def gbs(ms):
            return round(alg_launch / (ms * 1e-3) / 1e9, 3) if ms > 0 else None
        stages = {STAGES[i]: {"ms_per_call": round(mean_stage[i], 4), "achieved": gbs(mean_stage[i]), "frac": round(gbs(mean_stage[i]) / HBM_PEAK_GBS, 6) if mean_stage[i] > 0 else None}
                  for i in range(7)}
        stages["normals kernel (inside neighbours+normals)"] = {"ms_per_call": round(mean_stage[7], 4), "achieved": gbs(mean_stage[7]), "frac": round(gbs(mean_stage[7]) / HBM_PEAK_GBS, 6) if mean_stage[7] > 0 else None}
        roofline = {"bound": "hbm", "kernel": KERNELS[dom][0], "stage": KERNELS[dom][1], "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                    "launch_ms": round(dom_ms, 4), "frames_per_launch": round(frames_per_launch, 2),
                    "algorithmic_bytes_per_launch": int(alg_launch),
                    "dominant_by": "first kernel by total time in the rocprofv3 kernel trace of this command (profiles/, newest r*_kernel_stats.csv); HIP-event total of its launches in this run: %.1f ms" % stage_ms[dom],
                    "timer": "hipEventRecord pair around the one merge dispatch of every call, on the call's stream (f3ds_result.ms_stage[5]), mean over the timed region's %d calls" % calls_done[0],
                    "merge_layouts_last_call": {"%d waves, residency %d" % k: v for k, v in layouts.items()},
                    "whole_path": whole,
                    "stages": stages,
                    "stage_ms_per_call": {STAGES[i]: round(mean_stage[i], 4) for i in range(7)}}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from conftest import CpuChecker
            import hashlib
            # The baseline is timed on the libm-linked build of the oracle (glibc log / exp / cbrt / atan2 ...: what a real PCL / OpenCV build calls);
            # the shared-math build (csrc/f3ds_math.h, transcendentals from IEEE basic operations so that host and device agree to the bit) is
            # the parity checker and ~1.6x slower: its time is reported beside it, and the two label vectors must be identical.
            ora_sm = CpuChecker(os.path.join(ROOT, "oracle", "libf3ds_oracle.so"), "f3ds_oracle")
            ora = CpuChecker(os.path.join(ROOT, "oracle", "libf3ds_oracle_libm.so"), "f3ds_oracle")
            ts = time.perf_counter()
            rc_sm, olab_sm, _, oh_sm = ora_sm.segment(frames_host[0], prm)
            sm_s = time.perf_counter() - ts
            oh_sm.close()
            tc = time.perf_counter()
            rc, olab, ores, oh = ora.segment(frames_host[0], prm)
            cpu_s = time.perf_counter() - tc
            assert rc == 0 and rc_sm == 0
            libm_same = hashlib.sha256(olab.tobytes()).hexdigest() == hashlib.sha256(olab_sm.tobytes()).hexdigest()
            # what a user of the reference's main() waits for on top of the label path: refineSupervoxels(3) (viewer only) and a
            # second VCCS run on the label-coloured cloud (/root/reference/src/supervoxel_clustering.cpp:369-400)
            tm = time.perf_counter()
            oh.refine(3)
            p0 = prm.copy(); p0.threshold = 0.0
            rc2, _, _, oh2 = ora.segment(frames_host[1], p0)
            main_s = cpu_s + time.perf_counter() - tm
            oh2.close()
            cpu = cpu_baseline_record(npts, cpu_s, sm_s, main_s, libm_same, os.cpu_count())
            oh.close()
            # the fair multi-core number (SURVEY.md 8d iii): the 64-frame batch of the step, one frame per host core (the oracle is
            # single-threaded like the reference; ctypes releases the GIL, so these are real threads on real cores)
            ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            nthr = max(1, min(FPS, ncores))
            tp = time.perf_counter()
            with ThreadPoolExecutor(nthr) as ex:
                done = list(ex.map(lambda f: ora.segment(f, prm)[0], frames_host))
            par_s = time.perf_counter() - tp
            assert all(r == 0 for r in done)
            cpu["frames_parallel"] = {"value": round(FPS * npts / par_s / 1e6, 4) if libm_same else None, "unit": "Mpoints/s", "cores": nthr, "host_cpus": ncores, "seconds": round(par_s, 2),
                                      "sample": "the %d frames of one step, one frame per core on %d threads through oracle/libf3ds_oracle_libm.so (label path only)" % (FPS, nthr)}
        invalid = None
        what_if = {k: os.environ[k] for k in ("F3DS_BENCH_THRESHOLD", "F3DS_BENCH_PLAN", "F3DS_FAKE_MERGE", "F3DS_FAKE_MERGE_LDS") if os.environ.get(k)}
        if "+whatif" in lib_text:
            what_if["library"] = lib_text
        if "+dev" in lib_text:      # F3DS_DEV is set: the library reads its development switches (csrc/f3ds_dev.h) -- kernel layouts may differ from production
            what_if["F3DS_DEV"] = {k: v for k, v in os.environ.items() if k.startswith("F3DS_") and not k.startswith("F3DS_BENCH_")}
        if parity and (parity.get("error") or parity.get("mismatches_all_ranks")):
            invalid = "labels differ from the oracle's committed hashes" if not parity.get("error") else "label check failed: " + parity["error"]
        elif npts == 1000 * 1000 and not (parity and parity.get("frames_checked")):
            invalid = "no frame of the timed region was checked against the oracle's hashes"
        if npts != 1000 * 1000:
            what_if["frame"] = "%dx%d" % (args.width, args.height)
            invalid = None
        if what_if:      # a run that changes results or the measured configuration is a timing experiment, never a measurement
            invalid = "what-if run (%s): timing experiment, not a measurement%s" % (", ".join(sorted(what_if)), "; not the BASELINE workload (1000x1000-point frames)" if "frame" in what_if else "")
        what_if_value = value
        if invalid:
            value = None
        line = {"metric": "Mpoints/sec segmented end-to-end, 1M-pt RGB-D frames", "value": round(value, 3) if value is not None else None, "invalid": invalid, "unit": "Mpoints/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
                "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": ("step = ONE batch of %d distinct synthetic %dx%d (%d-point) XYZRGBA frames for the whole job, frame i on rank i mod N (BASELINE config 5 as written, seeds 1000..1063), "
                                        "-v 0.008 -s 0.08 --AL --CVX -t %g, frames resident in HBM" % (world * FPS, args.width, args.height, npts, prm.threshold)) if args.strong else
                                       ("step = batch of %d distinct synthetic %dx%d (%d-point) XYZRGBA frames per GPU (BASELINE config 5's batch, seeds 1000+64*rank..), "
                                        "-v 0.008 -s 0.08 --AL --CVX -t %g, frames resident in HBM" % (FPS, args.width, args.height, npts, prm.threshold)),
                           "frames_per_step_per_gpu": FPS, "points_per_step": world * FPS * npts, "frames_timed": world * total_frames,
                           "batch_calls": len(plan), "frames_per_call": round(total_frames / max(1, len(plan)), 1), "concurrent_calls": ngroups, "distinct_frames_per_gpu": FPS,
                           "setup": "two untimed passes over all contexts (scratch allocation up to the high-water marks of the 64 frames) before the warm-up steps",
                           "parallelism": ("%d ranks, one per GPU, %s" % (world, "frame i of every step on rank i mod N" if args.strong else "every rank its own frames")) if world > 1 else "1 GPU",
                           "label_gather": ("one RCCL gather of each step's [%d, 1M] label block (%d MB per rank) to rank 0, in step order, overlapped with later steps" % (FPS, FPS * 4)) if dist_on else "none (1 rank)",
                           "V": res.n_voxels, "S": res.n_supervoxels, "E": res.n_edges, "merges": res.n_merges, "regions": res.n_regions},
                "what_if": what_if or None, "what_if_value": round(what_if_value, 3) if what_if else None,
                "value_hbm_resident": round(value, 3) if value is not None else None,
                "value_survey_8d": host_io["value"] if host_io else None,
                "value_note": "`value`: frames resident in HBM, labels left in HBM (the bench contract: a PCIe-inclusive rate is never `value`); `value_survey_8d` = `value_host_io.value`: "
                              "pinned host buffer in -> labels in a pinned host buffer, the metric as SURVEY.md 8d words it",
                "value_host_io": host_io, "labels_checked": parity, "single_frame_latency_ms": round(latency_ms, 3), "config5_batch_latency_ms": batch_latency,
                "strong_scaling_estimate": strong_estimate(batch_latency, npts),
                "library": lib_text, "roofline": roofline, "cpu_baseline": cpu}
    for grp in ctxs:
        for c in grp:
            c.close()
    if dist_on:
        dist.destroy_process_group()
    if rank == 0:
        # the ONE JSON line, last on stdout: whatever C libraries have buffered on stdout (RCCL prints a version banner there) goes out first
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)
        if line.get("invalid"):
            sys.exit(1)


if __name__ == "__main__":
    main()
