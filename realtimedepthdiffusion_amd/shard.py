"""Host-side sharding of independent images over ranks: images never exchange data,
so the only collective anywhere near the path is the timing barrier / MAX-reduce used by bench.py.
Pure torch.distributed; works with the gloo backend on CPU (tests) and nccl (=RCCL) on GPUs."""


def images_for_rank(n_images, world, rank):
    """Round-robin: image i goes to rank i % world (the reference has a single device;
    /root/reference/src/main.cpp:116)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    return list(range(rank, n_images, world))


def fence(dist=None, device_sync=None):
    """Barrier bracketed by device syncs, as the bench contract asks."""
    if device_sync is not None:
        device_sync()
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
        if device_sync is not None:
            device_sync()


def aggregate(units_local, elapsed_local, dist=None, device="cpu"):
    """Whole-job throughput = units processed by ALL ranks / MAX over ranks of the elapsed time."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(units_local), float(elapsed_local), float(units_local) / float(elapsed_local)
    t = torch.tensor([float(elapsed_local)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units_local)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()), float(t.item()), float(u.item()) / float(t.item())
