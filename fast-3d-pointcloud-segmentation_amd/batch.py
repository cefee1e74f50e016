"""Frame sharding for multi-GPU batches (BASELINE.json config 5): independent frames, one process
per GPU, frame i -> rank i mod world; the only exchange is the label output, gathered to rank 0.

Kept free of GPU calls so the world_size-2 gloo test can drive it on CPU tensors; on the GPU box the
same functions run over RCCL (torch backend "nccl")."""


def frames_of_rank(n_frames, rank, world):
    """Indices of the frames rank `rank` segments (round-robin)."""
    return list(range(rank, n_frames, world))


def gather_labels(local_labels, dist, dst=0):
    """local_labels: list of 1-D int32 tensors (one per local frame, equal length across ranks for
    the synthetic batch).  Returns on `dst` a dict frame_index -> tensor, elsewhere None.  One
    gather per local frame slot: world x 4 MB for 1M-point frames, each peer over its own xGMI link."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    out = {} if rank == dst else None
    n_slots = torch.tensor([len(local_labels)], dtype=torch.int64, device=local_labels[0].device if local_labels else "cpu")
    slots = [torch.zeros_like(n_slots) for _ in range(world)]
    dist.all_gather(slots, n_slots)
    max_slots = int(max(int(s.item()) for s in slots))
    for k in range(max_slots):
        mine = local_labels[k] if k < len(local_labels) else torch.zeros_like(local_labels[0])
        bufs = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
        dist.gather(mine, bufs, dst=dst)
        if rank == dst:
            for r in range(world):
                if k < int(slots[r].item()):
                    out[r + k * world] = bufs[r]
    return out


def gather_label_block(block, dist, bufs=None, dst=0):
    """The form bench.py uses: the label output of a whole batch (a contiguous [frames, points] int32 block per rank)
    goes to rank `dst` in ONE collective.  `bufs` (rank dst only): preallocated list of world blocks, reused
    across batches.  Returns the list on dst, None elsewhere.  Collectives on one communicator must not be issued
    concurrently from several threads: callers with several batches in flight serialise this call."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    if rank == dst and bufs is None:
        bufs = [torch.empty_like(block) for _ in range(world)]
    dist.gather(block, bufs if rank == dst else None, dst=dst)
    return bufs if rank == dst else None
