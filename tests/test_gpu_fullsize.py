"""The north star's 4K (and 8K) Jacobi stencil sweep and BASELINE configs 3 and 5 at their full sizes, on the GPU, against
the oracle (through the C ABI).

North star: the reference's hot loop (src/GPUSolver.cu:295-309) at 3840x2160 x 1000 sweeps and 7680x4320 x 200 sweeps in the
library's DEFAULT configuration -- the very kernel instantiation `bench.py`'s `sweep_4k` record times -- bit for bit.

Config 3: 3840x2160, red-black Gauss-Seidel / SOR to a 1e-4 residual (`rtdd_solve_ex`, RTDD_METHOD_RED_BLACK_GS).
Config 5: 7680x4320, multigrid V-cycles (RTDD_METHOD_MULTIGRID).
Both are EXTENSIONS (the reference has one solver, src/GPUSolver.cu:274-316); what pins them is the oracle's restatement of
the same schedule, bit for bit where the oracle finishes in seconds, and oracle-recomputed residuals at the full length.
"""
import math

import numpy as np
import pytest

import realtimedepthdiffusion_amd as rt
from gpu_util import assert_bit_equal, down, up
from realtimedepthdiffusion_amd.synth import make_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def _ctx():
    c = rt.Context(0)
    c.GPULoadWeights(0.4)
    yield c
    c.close()


@pytest.fixture
def ctx(_ctx):
    _ctx.set_option(rt.OPT_FP_CONTRACT, 1); _ctx.set_option(rt.OPT_PERSISTENT, 1)
    for k in (rt.OPT_SWEEP_KERNEL, rt.OPT_TILE, rt.OPT_TEMPORAL_DEPTH, rt.OPT_ROWS_PER_WAVE):
        _ctx.set_option(k, 0)
    return _ctx


@pytest.fixture(scope="module")
def problem_4k(oracle):
    rows, cols = 2160, 3840
    p = make_problem(rows, cols, seed=1234)
    p["idx"] = oracle.index_to_weight(p["gray"], None, 0, 0)
    return p


@pytest.fixture(scope="module")
def problem_8k(oracle):
    rows, cols = 4320, 7680
    p = make_problem(rows, cols, seed=1234)
    p["idx"] = oracle.index_to_weight(p["gray"], None, 0, 0)
    return p


def sor_cycle_sweeps(rows, cols, cycles_needed):
    """Sweep count of rtdd_solve_ex's RTDD_RELAXATION_AUTO schedule (include/rtdd.h) when cycle `cycles_needed - 1` is the
    first to reach the tolerance in its k-th polish block: returned as the set of admissible totals."""
    longest = max(rows, cols)
    done, ok = 0, set()
    for cycle in range(cycles_needed):
        e = min(cycle, 6)
        done += (longest << e) + (longest << e) // 4
        for _k in range(5):
            done += 20
            if cycle == cycles_needed - 1:
                ok.add(done)
    return ok


def _jacobi_at_size(ctx, oracle, lut, p, iters, contract, what):
    rows, cols = p["gray"].shape
    ctx.GPUAllocateDeviceMemory(rows, cols, 1)
    ctx.set_option(rt.OPT_FP_CONTRACT, contract)
    want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], iters, 0, 0, lut, contract, threads=oracle.max_threads())
    d, m, g = up(p["depth"]), up(p["mask"]), up(p["gray"])
    ctx.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, iters, 1e-5, 0)
    info = ctx.last_solve_info()
    ctx.synchronize()
    got = down(d)
    assert info.kernel == 2 and info.iterations == iters, info.describe()          # the temporally blocked kernel, automatic tile / depth
    assert np.abs(got - want).max() <= 1e-4, f"{what}: {info.describe()}"
    assert_bit_equal(got, want, f"{what} [{info.describe()}]")
    dir_ = p["mask"] == 255
    assert np.array_equal(got[dir_], p["depth"][dir_])
    return info


@pytest.mark.parametrize("contract", [1, 0])
def test_north_star_4k_1000_jacobi_sweeps_bit_exact(ctx, oracle, lut, problem_4k, contract):
    """3840x2160 x 1000 Chebyshev-Jacobi sweeps, default options: what `sweep_4k` in the bench line measures."""
    info = _jacobi_at_size(ctx, oracle, lut, problem_4k, 1000, contract, "4K x 1000 Jacobi")
    assert info.persistent == 0 and info.launches == -(-1000 // info.temporal_depth), info.describe()   # too many tiles to be resident: launch per block


def test_north_star_4k_odd_sweep_count_and_second_seed(ctx, oracle, lut):
    """Another image and a sweep count that is no multiple of the temporal depth (a short last block)."""
    p = make_problem(2160, 3840, seed=77)
    _jacobi_at_size(ctx, oracle, lut, p, 203, 1, "4K x 203 Jacobi, seed 77")


@pytest.mark.parametrize("tile,depth", [(6, 8), (5, 8), (7, 12), (8, 8), (10, 12), (4, 8)])
def test_4k_jacobi_every_large_tile_bit_exact(ctx, oracle, lut, problem_4k, tile, depth):
    """The tile shapes the cost model can pick for large images, each forced at 4K for 64 sweeps."""
    p = problem_4k
    rows, cols = p["gray"].shape
    ctx.GPUAllocateDeviceMemory(rows, cols, 1)
    want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 64, 0, 0, lut, 1, threads=oracle.max_threads())
    ctx.set_option(rt.OPT_SWEEP_KERNEL, 2); ctx.set_option(rt.OPT_TILE, tile); ctx.set_option(rt.OPT_TEMPORAL_DEPTH, depth)
    d, m, g = up(p["depth"]), up(p["mask"]), up(p["gray"])
    ctx.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 64, 1e-5, 0)
    info = ctx.last_solve_info()
    ctx.synchronize()
    assert info.tile == tile, info.describe()      # (temporal_depth reports the LAST launch: the short tail block, or the tile's cap)
    assert_bit_equal(down(d), want, f"4K x 64, tile {tile} depth {depth}")


def test_north_star_8k_200_jacobi_sweeps_bit_exact(ctx, oracle, lut, problem_8k):
    """7680x4320 x 200 sweeps, default options (the HBM-resident size: 564 MB at 17 B/px)."""
    _jacobi_at_size(ctx, oracle, lut, problem_8k, 200, 1, "8K x 200 Jacobi")


@pytest.mark.parametrize("contract,omega", [(1, 1.9), (0, 1.0)])
def test_config3_4k_red_black_64_sweeps_bit_exact(ctx, oracle, lut, problem_4k, contract, omega):
    """3840x2160, 64 red-black sweeps (launch-per-block path: 4K has more tiles than CUs), every pixel == the oracle's sweep."""
    p = problem_4k
    rows, cols = p["gray"].shape
    ctx.GPUAllocateDeviceMemory(rows, cols, 1)
    ctx.set_option(rt.OPT_FP_CONTRACT, contract)
    x = p["depth"].copy()
    oracle.rbgs_sweeps_mt(x, p["idx"], p["mask"], lut, contract, omega, 64)
    d, m, g = up(p["depth"]), up(p["mask"]), up(p["gray"])
    its, _ = ctx.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=64, tolerance=0.0, relaxation=omega)
    assert its == 64
    assert_bit_equal(down(d), x, f"4K red-black x64, contract {contract}, omega {omega}")


def test_config3_4k_sor_cycles_to_1e_4(ctx, oracle, lut, problem_4k):
    """BASELINE config 3 itself: 4K from the cold start to max|J(x)-x| <= 1e-4 by SOR cycles.  The sweep count is one the
    schedule can stop at, the reported residual is the oracle's residual of the returned image (same bits), Dirichlet
    pixels are untouched and the result lies in the label hull."""
    p = problem_4k
    rows, cols = p["gray"].shape
    ctx.GPUAllocateDeviceMemory(rows, cols, 1)
    d, m, g = up(p["depth"]), up(p["mask"]), up(p["gray"])
    its, res = ctx.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=400000, tolerance=1e-4, relaxation=rt.RELAXATION_AUTO)
    got = down(d)
    assert res <= 1e-4
    assert any(its in sor_cycle_sweeps(rows, cols, c) for c in (1, 2, 3)), its
    assert np.float32(oracle.residual_mt(got, p["idx"], p["mask"], lut, 1)) == np.float32(res)
    dir_ = p["mask"] == 255
    assert np.array_equal(got[dir_], p["depth"][dir_])
    assert got.min() >= 0.0 and got.max() <= 255.0          # every red-black update is clamped (include/rtdd.h)
    # the tail of the schedule restated: 20 more plain sweeps on both sides stay bit-identical and keep the residual under the tolerance
    x = got.copy()
    oracle.rbgs_sweeps_mt(x, p["idx"], p["mask"], lut, 1, 1.0, 20)
    d2 = up(got)
    ctx.solve_ex(d2, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=20, tolerance=0.0, relaxation=1.0)
    assert_bit_equal(down(d2), x, "20 Gauss-Seidel sweeps from the converged 4K image")
    assert oracle.residual_mt(x, p["idx"], p["mask"], lut, 1) <= 1e-4


def test_config5_8k_multigrid_two_cycles_bit_exact(ctx, oracle, lut, problem_8k):
    """7680x4320: every plane of every level of the hierarchy and the iterate after two V-cycles equal the restatement's."""
    p = problem_8k
    rows, cols = p["gray"].shape
    ctx.GPUAllocateDeviceMemory(rows, cols, 1)
    x = p["depth"].copy()
    _, _, nlev = oracle.mg_solve(x, p["idx"], p["mask"], lut, 1, 2, 0.0, 1)
    d, m, g = up(p["depth"]), up(p["mask"]), up(p["gray"])
    its, _ = ctx.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_MULTIGRID, maxIterations=2, tolerance=0.0)
    assert its == 2 and nlev >= 7
    assert_bit_equal(down(d), x, "8K iterate after 2 cycles")
    for lvl in range(nlev):
        for which in range(9 if lvl + 1 < nlev else 5):
            assert_bit_equal(ctx.multigrid_level(lvl, which), oracle.mg_level(lvl, which), f"8K level {lvl} plane {which}")


def test_config5_8k_multigrid_to_1e_4(ctx, oracle, lut, problem_8k):
    """BASELINE config 5 itself: 8K from the cold start to a 1e-4 residual by V-cycles; the reported residual is the
    oracle's residual of the returned image, and cycle count, residual and every pixel equal the restated driver's."""
    p = problem_8k
    rows, cols = p["gray"].shape
    ctx.GPUAllocateDeviceMemory(rows, cols, 1)
    d, m, g = up(p["depth"]), up(p["mask"]), up(p["gray"])
    its, res = ctx.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_MULTIGRID, maxIterations=60, tolerance=1e-4)
    got = down(d)
    assert res <= 1e-4 and its <= 30, (its, res)
    assert np.float32(oracle.residual_mt(got, p["idx"], p["mask"], lut, 1)) == np.float32(res)
    dir_ = p["mask"] == 255
    assert np.array_equal(got[dir_], p["depth"][dir_])
    x = p["depth"].copy()
    want_its, want_res, _ = oracle.mg_solve(x, p["idx"], p["mask"], lut, 1, 60, 1e-4, 1)
    assert (its, np.float32(res)) == (want_its, np.float32(want_res))
    assert_bit_equal(got, x, "8K residual-stopped V-cycles")
