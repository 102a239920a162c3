"""bench.py's N > 1 path: `python3 bench.py --gpus N` started plainly is its own launcher (one process per rank, 127.0.0.1
rendezvous, gloo or RCCL for the timing barrier only), deals a fixed batch round-robin, and rank 0 prints ONE JSON line.
On a CPU-only box --dry-run runs all of that except the GPU work; with a GPU two ranks share device 0."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_plain_launch_two_ranks_dry_run_batch():
    d = _run(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", "--workload", "batch64_1080p"])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "strong"
    assert d["config"]["images_per_step"] == 64 and d["config"]["images_this_rank"] == 32
    assert "configs[3]" in d["config"]["workload"] and d["unit"] == "Mpixel-iterations/s"
    # SUM of the units of both ranks / MAX of their times: 3 steps x 64 images x 1080p x 1000 sweeps
    assert abs(d["value"] * 1e6 * d["ms_per_step"] * 1e-3 * 3 - 3 * 64 * 1080 * 1920 * 1000) < 1e-3 * 3 * 64 * 1080 * 1920 * 1000


def test_plain_launch_default_workload_is_weak():
    d = _run(["--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["images_per_step"] == 2 and d["config"]["images_this_rank"] == 1
    d1 = _run(["--gpus", "1", "--dry-run", "--steps", "2", "--warmup", "1"])
    assert d1["n_gpus"] == 1 and d1["config"]["images_per_step"] == 1


def test_eight_ranks_dry_run_both_scalings():
    """What the driver's scaling run launches, minus the GPUs: eight ranks, the default (one image per rank: weak) and BASELINE
    configs[3] (64 images dealt to 8 ranks: strong); SUM of units / MAX of time on rank 0, one JSON line."""
    d = _run(["--gpus", "8", "--dry-run", "--steps", "2", "--warmup", "1"], timeout=600)
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["images_per_step"] == 8 and d["config"]["images_this_rank"] == 1
    d = _run(["--gpus", "8", "--dry-run", "--steps", "2", "--warmup", "1", "--workload", "batch64_1080p"], timeout=600)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["images_per_step"] == 64 and d["config"]["images_this_rank"] == 8


def test_batched_estimates_shard_like_the_batch_of_solves():
    """`batch64_1080p_estimate`: the same 64 images as whole estimates, image i on rank i % N, a rank's images as one
    batched pyramid; units = the pixel-sweeps of the five levels of every image."""
    per_estimate = sum((1080 >> l) * (1920 >> l) * int(1000 / 2 ** (4 - l)) for l in range(5))
    for n, mine in ((2, 32), (8, 8)):
        d = _run(["--gpus", str(n), "--dry-run", "--steps", "2", "--warmup", "1", "--workload", "batch64_1080p_estimate"], timeout=600)
        assert d["n_gpus"] == n and d["scaling"] == "strong" and d["config"]["images_per_step"] == 64 and d["config"]["images_this_rank"] == mine
        assert "estimates" in d["config"]["workload"]
        assert abs(d["value"] * 1e6 * d["ms_per_step"] * 1e-3 * 2 - 2 * 64 * per_estimate) < 1e-3 * 2 * 64 * per_estimate


def test_gpus_flag_must_match_world_size():
    e = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--dry-run"], env=e, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=2" in (out.stderr + out.stdout)


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_for_real():
    """The product through the launcher: two ranks on device 0 (RTDD_BENCH_SHARE_GPU), a small image each."""
    d = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--workload", "270x480x100"], env={"RTDD_BENCH_SHARE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["config"]["images_per_step"] == 2 and d["value"] > 0
    assert d["config"]["persistent"] == 0                      # ranks sharing a GPU never launch persistently
    assert d["roofline"]["bound"] == "valu" and 0 < d["roofline"]["frac"] < 1


@pytest.mark.gpu
def test_two_ranks_results_equal_the_oracle():
    """Two ranks through the launcher, three images each (a fixed batch of 6 dealt round-robin), and every rank's depth maps
    compared with the oracle bit for bit (--verify): the product under N > 1, not just its plumbing."""
    import bench
    bench.WORKLOADS  # noqa: B018  (the module imports without touching the GPU)
    d = _run(["--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--workload", "batch6_540x960x120", "--verify"], env={"RTDD_BENCH_SHARE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["images_per_step"] == 6 and d["config"]["images_this_rank"] == 3
    v = d["verified"]
    assert v["images_differing_all_ranks"] == 0 and v["rank0_images"] == [0, 2, 4] and "bit for bit" in v["against"], v


@pytest.mark.gpu
def test_rccl_that_cannot_come_up_falls_back_to_gloo_on_all_ranks():
    """Two ranks on ONE device cannot form an RCCL communicator (duplicate GPU).  Asked for RCCL anyway, every rank must end up on gloo
    -- decided collectively (shard.timing_group), never rank by rank -- and the run must complete with verified results."""
    d = _run(["--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--workload", "batch4_270x480x60", "--verify"],
             env={"RTDD_BENCH_SHARE_GPU": "1", "RTDD_BENCH_BACKEND": "nccl"}, timeout=600)
    assert d["n_gpus"] == 2 and d["config"]["timing_barrier_backend"] == "gloo"
    assert d["verified"]["images_differing_all_ranks"] == 0


@pytest.mark.gpu
def test_config4_batch64_on_one_gpu_verified():
    """BASELINE configs[3] at N = 1 (the base of the scaling curve): 64 independent 1080p images x 1000 sweeps on one GPU, one
    stream, every one of the 64 depth maps equal to the oracle's (--verify)."""
    d = _run(["--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--workload", "batch64_1080p", "--verify"], timeout=900)
    assert d["n_gpus"] == 1 and d["config"]["images_per_step"] == 64 and d["config"]["images_this_rank"] == 64 and d["scaling"] == "strong"
    assert "configs[3]" in d["config"]["workload"]
    assert d["verified"]["images_differing_all_ranks"] == 0 and len(d["verified"]["rank0_images"]) == 64
    assert d["config"]["persistent"] == 1 and d["value"] > 0


@pytest.mark.gpu
def test_two_ranks_run_their_share_of_a_batch_as_batched_estimates():
    """Two ranks through the launcher, each its three images of a batch of six as ONE batched pyramid (rtdd_estimate_depth_batch)."""
    d = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--workload", "batch6_540x960x400_estimate", "--verify"], env={"RTDD_BENCH_SHARE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["images_per_step"] == 6 and d["config"]["images_this_rank"] == 3
    assert d["value"] > 0 and d["estimates_per_s"] > 0
    v = d["verified"]               # every rank's three images, all levels + the u8 map, cold and warm-started, against the oracle's cascade
    assert v["images_differing_all_ranks"] == 0 and v["images_checked_all_ranks"] == 6 and v["rank0_images_checked"] == [0, 2, 4], v


@pytest.mark.gpu
def test_the_batched_estimate_workload_the_bench_line_quotes_is_verified():
    """`batch64_1080p_estimate` at N = 1 with --verify: the timed batch's sampled images (0, 1, 7, 8, 31, 32, 62, 63) against the oracle's
    cascade level by level, and the per-level choices of the timed batch reported next to the figure."""
    d = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--workload", "batch64_1080p_estimate", "--verify"], timeout=900)
    v = d["verified"]
    assert v["images_differing_all_ranks"] == 0 and v["rank0_images_checked"] == [0, 1, 7, 8, 31, 32, 62, 63] and v["levels"] == 5, v
    assert v["rank0_level_choices"]["0"]["images_per_launch"] == 1 and v["rank0_level_choices"]["4"]["images_per_launch"] == 64
    assert d["estimates_per_s"] > 0
