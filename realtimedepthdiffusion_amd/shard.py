"""Host-side sharding of independent images over ranks: images never exchange data,
so the only collective anywhere near the path is the timing barrier / MAX-reduce used by bench.py.
Pure torch.distributed; works with the gloo backend on CPU (tests) and nccl (=RCCL) on GPUs."""


def images_for_rank(n_images, world, rank):
    """Round-robin: image i goes to rank i % world (the reference has a single device;
    /root/reference/src/main.cpp:116)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    return list(range(rank, n_images, world))


def fence(dist=None, device_sync=None, group=None):
    """Barrier bracketed by device syncs, as the bench contract asks.  `group`: the process group to use (None = the default one)."""
    if device_sync is not None:
        device_sync()
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier(group=group)
        if device_sync is not None:
            device_sync()


def aggregate(units_local, elapsed_local, dist=None, device="cpu", group=None):
    """Whole-job throughput = units processed by ALL ranks / MAX over ranks of the elapsed time."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(units_local), float(elapsed_local), float(units_local) / float(elapsed_local)
    t = torch.tensor([float(elapsed_local)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units_local)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(u, op=dist.ReduceOp.SUM, group=group)
    return float(u.item()), float(t.item()), float(u.item()) / float(t.item())


def timing_group(dist, device, want_rccl, timeout_s=60):
    """The process group for the timing barrier and the two scalar reductions (nothing else crosses ranks).  Every rank is already
    in a gloo default group (it always comes up).  If RCCL is wanted, all ranks create an RCCL group and try one all-reduce on it;
    whether that worked is agreed over gloo (MIN of a flag), so either ALL ranks use RCCL or ALL stay on gloo -- a rank-by-rank
    fallback could leave the ranks on different backends, waiting at the first barrier.  The RCCL probe's timeout (60 s) is well below the
    gloo default group's (bench.py: 300 s), so a rank whose RCCL group failed at once still finds the others at the vote when their
    probe has timed out.  A group that lost the vote is destroyed (best effort).  Returns (group or None, "nccl" | "gloo")."""
    import datetime
    import torch
    if not want_rccl:
        return None, "gloo"
    ok, g = 1, None
    try:
        g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=timeout_s))
        probe = torch.ones(1, device=device)
        dist.all_reduce(probe, group=g)
        torch.cuda.synchronize()
        ok = int(abs(float(probe.item()) - dist.get_world_size()) < 0.5)
    except Exception:                                               # noqa: BLE001
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)                     # the default (gloo) group
    if int(flag.item()) == 1:
        return g, "nccl"
    if g is not None:
        try:
            dist.destroy_process_group(g)
        except Exception:                                           # noqa: BLE001
            pass
    return None, "gloo"
