"""Frame sharding for multi-GPU batches (BASELINE.json config 5): independent frames, one process
per GPU; the only exchange is the label output, gathered to rank 0 -- one collective per 64-frame step.

Kept free of GPU calls so the world_size-2 gloo tests can drive it on CPU tensors; on the GPU box the
same functions run over RCCL (torch backend "nccl").  The C++ counterpart (one host thread per GPU, labels
to GPU 0 with librccl) is csrc/f3ds_multi.cpp."""
import threading


def frames_of_rank(n_frames, rank, world):
    """Indices of the frames rank `rank` segments (round-robin)."""
    return list(range(rank, n_frames, world))


def gather_labels(local_labels, dist, dst=0, n_points=None, device="cpu"):
    """local_labels: list of 1-D int32 tensors (one per local frame, equal length across ranks for
    the synthetic batch).  Returns on `dst` a dict frame_index -> tensor, elsewhere None.  One
    gather per local frame slot: world x 4 MB for 1M-point frames, each peer over its own xGMI link.
    A rank may hold no frame at all (fewer frames than ranks): it pads with zeros of the common length."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    out = {} if rank == dst else None
    if local_labels:
        device = local_labels[0].device
        n_points = int(local_labels[0].numel())
    meta = torch.tensor([len(local_labels), n_points or 0], dtype=torch.int64, device=device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    slots = [int(m[0].item()) for m in metas]
    length = max(int(m[1].item()) for m in metas)
    for k in range(max(slots)):
        mine = local_labels[k] if k < len(local_labels) else torch.zeros(length, dtype=torch.int32, device=device)
        bufs = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
        dist.gather(mine, bufs, dst=dst)
        if rank == dst:
            for r in range(world):
                if k < slots[r]:
                    out[r + k * world] = bufs[r]
    return out


def gather_label_block(block, dist, bufs=None, dst=0):
    """The label output of one step (a contiguous [frames, points] int32 block per rank) goes to rank `dst` in ONE
    collective.  `bufs` (rank dst only): preallocated list of world blocks, reused across steps.  Returns the list on
    dst, None elsewhere.  Collectives on one communicator must be issued in the same order on every rank and never
    concurrently from several threads: StepPipeline below owns that."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    if rank == dst and bufs is None:
        bufs = [torch.empty_like(block) for _ in range(world)]
    dist.gather(block, bufs if rank == dst else None, dst=dst)
    return bufs if rank == dst else None


def assemble_strong(bufs):
    """Strong scaling (bench.py --strong, BASELINE config 5 as written): a step is ONE batch whose frame i ran on rank i mod N, so after
    the step's gather block r holds the global frames r, r + N, r + 2N, ...  Returns the [frames, points] block in global frame order."""
    import torch
    world = len(bufs)
    out = torch.empty((world * bufs[0].shape[0],) + tuple(bufs[0].shape[1:]), dtype=bufs[0].dtype, device=bufs[0].device)
    for r, b in enumerate(bufs):
        out[r::world] = b
    return out


def plan_batches(total_frames, max_batch, groups, ramp=False):
    """Cut `total_frames` consecutive frames into batch calls of at most `max_batch` frames.  Returns [(first, last+1), ...].

    ramp=False: the number of calls is a multiple of `groups` (no host thread idles in the last round) and the calls are as
    equal as possible.
    ramp=True (what bench.py uses): the first `groups` calls grow (max_batch/groups, 2 max_batch/groups, ...) and the last
    `groups` shrink the same way, equal calls in between.  The host threads then reach the merge stage -- one long
    dispatch that holds most CUs -- at different times from the first round on, so that one call's wide kernels always
    have another call's merge loops to overlap with, and they finish together instead of leaving the last call alone on
    the GPU.  Short runs (the driver's --steps 20) spend most of their time in these two rounds."""
    if total_frames <= 0:
        return []
    max_batch, groups = max(1, max_batch), max(1, groups)
    import os
    forced = os.environ.get("F3DS_BENCH_PLAN")      # (development, bench.py voids the run: call sizes as "30,60,...,*,...,30" -- "*" = equal calls of <= max_batch for the rest)
    if forced:
        parts = forced.split(",")
        fixed = sum(int(x) for x in parts if x != "*")
        rest = total_frames - fixed
        sizes = []
        for x in parts:
            if x != "*":
                sizes.append(int(x))
            elif rest > 0:
                calls = -(-rest // max_batch)
                size, extra = divmod(rest, calls)
                sizes += [size + (1 if i < extra else 0) for i in range(calls)]
        if sum(sizes) == total_frames:
            out, f = [], 0
            for k in sizes:
                out.append((f, f + k)); f += k
            return out
    sizes = []
    if ramp and groups > 1 and total_frames >= 2 * max_batch:
        up = [max(1, -(-max_batch * (g + 1) // groups)) for g in range(groups)]
        head, tail = list(up), list(reversed(up))
        if sum(head) + sum(tail) > total_frames:      # not enough frames for two full ramps: scale both
            f = total_frames / float(sum(head) + sum(tail))
            head = [max(1, int(x * f)) for x in head]; tail = [max(1, int(x * f)) for x in tail]
        mid = total_frames - sum(head) - sum(tail)
        if mid < 0:
            return plan_batches(total_frames, max_batch, groups, False)
        while 0 < mid < max(2, max_batch // groups):  # a remainder too small for a call of its own goes to the big calls
            for lst, i in ((head, -1), (tail, 0), (head, -2), (tail, 1)):
                if mid > 0 and lst[i] < max_batch:
                    lst[i] += 1; mid -= 1
            if head[-1] >= max_batch and tail[0] >= max_batch and head[-2] >= max_batch and tail[1] >= max_batch:
                break
        calls = -(-mid // max_batch) if mid > 0 else 0
        size, extra = divmod(mid, calls) if calls else (0, 0)
        sizes = head + [size + (1 if i < extra else 0) for i in range(calls)] + tail
    else:
        calls = -(-total_frames // max_batch)
        if total_frames >= groups:
            calls = -(-calls // groups) * groups
        size, extra = divmod(total_frames, calls)
        sizes = [size + (1 if i < extra else 0) for i in range(calls)]
    out, f = [], 0
    for k in sizes:
        if k > 0:
            out.append((f, f + k))
            f += k
    assert f == total_frames
    return out


class StepPipeline:
    """K steps of `frames_per_step` independent frames on one rank.  `groups` host threads run batch calls (frame ranges
    from plan_batches, which need not respect step boundaries) through `run_batch(group, first, last)`; whenever every
    frame of step s is done -- in step order, the same order on every rank -- `on_step(s)` runs on a dedicated thread
    (the label gather of that step).  A step's label block is one of `n_blocks` ring slots: a batch that would write
    into the slot of a step not gathered yet waits.  Exceptions of any thread are re-raised by run()."""

    def __init__(self, frames_per_step, max_batch, groups, n_blocks, run_batch, on_step=None, ramp=False):
        self.fps, self.max_batch, self.groups, self.n_blocks, self.ramp = frames_per_step, max_batch, groups, n_blocks, ramp
        self.run_batch, self.on_step = run_batch, on_step
        if on_step is not None and n_blocks * frames_per_step < max_batch + frames_per_step:
            raise ValueError("label ring too small for one batch call")

    def block_of(self, frame):
        return (frame // self.fps) % self.n_blocks, frame % self.fps

    def run(self, n_steps):
        total = n_steps * self.fps
        plan = plan_batches(total, self.max_batch, self.groups, self.ramp)
        cv = threading.Condition()
        state = {"next": 0, "done": [0] * n_steps, "gathered": 0, "err": None}

        def worker(g):
            try:
                while True:
                    with cv:
                        if state["err"] or state["next"] >= len(plan):
                            return
                        f0, f1 = plan[state["next"]]
                        state["next"] += 1
                        if self.on_step is not None:      # ring slot free again?
                            last_step = (f1 - 1) // self.fps
                            while state["gathered"] < last_step - self.n_blocks + 1 and not state["err"]:
                                cv.wait()
                            if state["err"]:
                                return
                    self.run_batch(g, f0, f1)
                    with cv:
                        for s in range(f0 // self.fps, (f1 - 1) // self.fps + 1):
                            lo, hi = max(f0, s * self.fps), min(f1, (s + 1) * self.fps)
                            state["done"][s] += hi - lo
                        cv.notify_all()
            except BaseException as e:      # noqa
                with cv:
                    state["err"] = state["err"] or e
                    cv.notify_all()

        def gatherer():
            try:
                for s in range(n_steps):
                    with cv:
                        while state["done"][s] < self.fps and not state["err"]:
                            cv.wait()
                        if state["err"]:
                            return
                    self.on_step(s)
                    with cv:
                        state["gathered"] = s + 1
                        cv.notify_all()
            except BaseException as e:      # noqa
                with cv:
                    state["err"] = state["err"] or e
                    cv.notify_all()

        threads = [threading.Thread(target=worker, args=(g,)) for g in range(self.groups)]
        if self.on_step is not None:
            threads.append(threading.Thread(target=gatherer))
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if state["err"]:
            raise state["err"]
        return plan
