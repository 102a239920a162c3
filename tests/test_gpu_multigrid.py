"""Multigrid V-cycle (csrc/multigrid.hip; EXTENSION, BASELINE config 5) against its CPU restatement, through the C ABI."""
import numpy as np
import pytest

import realtimedepthdiffusion_amd as rt
from gpu_util import assert_bit_equal, down, up
from golden_util import LEVELS, NAMES, load
from realtimedepthdiffusion_amd.synth import make_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = rt.Context(0)
    c.GPULoadWeights(0.4)
    yield c
    c.close()


@pytest.mark.parametrize("rows,cols,seed,cycles", [(96, 128, 12, 3), (75, 133, 4, 2), (200, 150, 3, 2), (270, 480, 1234, 2), (17, 300, 8, 2), (16, 16, 2, 2), (33, 7, 1, 2)])
@pytest.mark.parametrize("contract", [1, 0])
def test_hierarchy_and_cycles_bit_exact(ctx, oracle, lut, rows, cols, seed, cycles, contract):
    """Every plane of every level of the hierarchy (couplings, diagonal, interpolation weights) and the iterate after a
    few V-cycles equal the restatement's bit for bit (one thread per point, fixed evaluation order on both sides)."""
    p = make_problem(rows, cols, seed=seed)
    ctx.GPUAllocateDeviceMemory(rows, cols, 1)
    ctx.set_option(rt.OPT_FP_CONTRACT, contract)
    idx = oracle.index_to_weight(p["gray"], None, 0, 0)
    x = p["depth"].copy()
    _, _, nlev = oracle.mg_solve(x, idx, p["mask"], lut, contract, cycles, 0.0, 1)
    d, m, g = up(p["depth"]), up(p["mask"]), up(p["gray"])
    its, _ = ctx.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_MULTIGRID, maxIterations=cycles, tolerance=0.0)
    assert its == cycles
    for lvl in range(nlev):
        for which in range(9 if lvl + 1 < nlev else 5):
            assert_bit_equal(ctx.multigrid_level(lvl, which), oracle.mg_level(lvl, which), f"level {lvl} plane {which}")
    assert_bit_equal(down(d), x, f"iterate after {cycles} cycles")
    ctx.set_option(rt.OPT_FP_CONTRACT, 1)


def test_reaches_the_residual_and_reports_it(ctx, oracle, lut):
    """Full size (BASELINE config 2's image, config 5's method): V-cycles from the cold start to a 1e-4 residual, the
    same cycle count, reported residual and bits as the restatement."""
    rows, cols = 1080, 1920
    p = make_problem(rows, cols, seed=1234)
    ctx.GPUAllocateDeviceMemory(rows, cols, 1)
    idx = oracle.index_to_weight(p["gray"], None, 0, 0)
    d, m, g = up(p["depth"]), up(p["mask"]), up(p["gray"])
    its, res = ctx.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_MULTIGRID, maxIterations=40, tolerance=1e-4)
    got = down(d)
    assert res <= 1e-4 and its <= 25, (its, res)
    assert oracle.residual(got, idx, p["mask"], lut, 1) == np.float32(res)
    x = p["depth"].copy()
    want_its, want_res, _ = oracle.mg_solve(x, idx, p["mask"], lut, 1, 40, 1e-4, 1)
    assert (its, res) == (want_its, np.float32(want_res))
    assert_bit_equal(got, x, "residual-stopped V-cycles")


@pytest.mark.parametrize("name", NAMES)
def test_agrees_with_scipy_direct_solution(ctx, name):
    g = load(name)
    lvl = LEVELS - 1
    r = 256 >> lvl
    ctx.GPUAllocateDeviceMemory(r, r, 1)
    d = up(g[f"depth_before_c1_L{lvl}"])
    its, res = ctx.solve_ex(d, up(g[f"mask{lvl}"]), up(g["direct_gray_L2"]), r, r, 0, method=rt.METHOD_MULTIGRID, maxIterations=10, tolerance=1e-30, checkEvery=10)
    assert its == 10 and res <= 5e-5
    assert np.abs(down(d) - g["direct_solution_L2"]).max() < 2e-4


@pytest.mark.parametrize("rows,cols,seed", [(135, 240, 1234), (270, 480, 1234), (540, 960, 77)])
def test_auto_method_vcycles_then_sor_cycles(ctx, oracle, lut, rows, cols, seed):
    """RTDD_METHOD_AUTO: V-cycles until the tolerance or until the remaining cycles are modelled dearer than SOR cycles, then those.
    Small images (a cycle is launch-bound: the model switches early) and one where the cycles stall on a thin strip; cycle count, sweep count, residual and bits as the same logic driven through the restatements."""
    from test_gpu_parity import _sor_cycles_restated
    p = make_problem(rows, cols, seed=seed)
    ctx.GPUAllocateDeviceMemory(rows, cols, 1)
    idx = oracle.index_to_weight(p["gray"], None, 0, 0)
    d, m, g = up(p["depth"]), up(p["mask"]), up(p["gray"])
    its, res = ctx.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_AUTO, maxIterations=200000, tolerance=1e-4)
    cycles = ctx.last_cycles
    x = p["depth"].copy()
    px = rows * cols
    sor_seconds = (((max(rows, cols) + 1) // 2) * 1.25 + 20.0) * max(px / 700e9, 2.5e-6)       # csrc/api.cpp
    want_cycles, want_res, _ = oracle.mg_solve(x, idx, p["mask"], lut, 1, 60, 1e-4, 1, alternative_seconds=sor_seconds)
    want_its = 0
    if want_res > 1e-4:
        want_its, want_res = _sor_cycles_restated(oracle, x, idx, p["mask"], lut, 1, 1e-4, 200000, halve=True)
    assert (cycles, its, res) == (want_cycles, want_its, np.float32(want_res)) and res <= 1e-4
    assert cycles >= 3 and its > 0            # the rule needs two earlier residuals; at these sizes the sweeps always finish
    assert_bit_equal(down(d), x, "auto")
