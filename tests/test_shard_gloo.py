"""The N > 1 PLUMBING on CPU: two gloo ranks deal a batch of independent images round-robin, the SUM / MAX aggregation of
bench.py and the barrier work, and nothing crosses ranks on the data path.  There is no GPU here and the product has no CPU
fallback, so the per-image work is stood in for by the oracle -- this file checks the sharding code, not the solver.  The
PRODUCT under two ranks is checked on the GPU box: tests/test_bench_launcher.py::test_two_ranks_results_equal_the_oracle
(every rank's depth maps against the oracle, bit for bit)."""
import os
import socket

import numpy as np
import pytest

from realtimedepthdiffusion_amd import shard


def test_round_robin_partition_is_disjoint_and_complete():
    for n, w in ((64, 8), (7, 2), (3, 4), (0, 2)):
        parts = [shard.images_for_rank(n, w, r) for r in range(w)]
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        shard.images_for_rank(4, 2, 2)


def _worker(rank, world, port, n_images, out_dir):
    import time
    import torch.distributed as dist
    import oracle
    from realtimedepthdiffusion_amd.synth import make_problem
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lut = oracle.load_weights(0.4)
    mine = shard.images_for_rank(n_images, world, rank)
    shard.fence(dist)
    t0 = time.perf_counter()
    for i in mine:
        p = make_problem(24, 32, seed=1234 + i)
        d = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 20, 0, 0, lut, 1)
        np.save(os.path.join(out_dir, f"img{i}.npy"), d)
    shard.fence(dist)
    el = time.perf_counter() - t0 + 0.01 * (rank + 1)        # make the ranks' times differ on purpose
    units, tmax, thr = shard.aggregate(len(mine) * 24 * 32 * 20, el, dist)
    if rank == 0:
        np.save(os.path.join(out_dir, "agg.npy"), np.array([units, tmax, thr, el]))
    else:
        np.save(os.path.join(out_dir, "el1.npy"), np.array([el]))
    dist.destroy_process_group()


def test_two_rank_gloo_batch_matches_single_process(tmp_path, oracle, lut):
    import torch.multiprocessing as mp
    from realtimedepthdiffusion_amd.synth import make_problem
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    n_images = 5
    mp.spawn(_worker, args=(2, port, n_images, str(tmp_path)), nprocs=2, join=True)
    for i in range(n_images):          # per-image results identical to a single-process run, bit for bit
        p = make_problem(24, 32, seed=1234 + i)
        want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 20, 0, 0, lut, 1)
        assert np.array_equal(np.load(tmp_path / f"img{i}.npy"), want)
    units, tmax, thr, el0 = np.load(tmp_path / "agg.npy")
    el1 = float(np.load(tmp_path / "el1.npy")[0])
    assert units == n_images * 24 * 32 * 20                  # SUM over ranks
    assert tmax == max(el0, el1) and abs(thr - units / tmax) < 1e-9      # MAX over ranks
