"""Multigrid V-cycle (csrc/multigrid.hip; EXTENSION, BASELINE config 5) against its CPU restatement, through the C ABI."""
import numpy as np
import pytest

import realtimedepthdiffusion_amd as rt
from gpu_util import assert_bit_equal, down, up
from golden_util import LEVELS, NAMES, load
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
    """The shared context with every option back at its default."""
    _ctx.set_option(rt.OPT_FP_CONTRACT, 1); _ctx.set_option(rt.OPT_PERSISTENT, 1)
    for k in (rt.OPT_SWEEP_KERNEL, rt.OPT_TILE, rt.OPT_TEMPORAL_DEPTH, rt.OPT_ROWS_PER_WAVE):
        _ctx.set_option(k, 0)
    return _ctx


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
    sor_seconds, cycle_seconds = ctx.auto_model(rows, cols)          # the library's own constants (RTDD_OPT_AUTO_*), not a copy
    want_cycles, want_res, _ = oracle.mg_solve(x, idx, p["mask"], lut, 1, 60, 1e-4, 1, alternative_seconds=sor_seconds, cycle_seconds=cycle_seconds)
    want_its = 0
    if want_res > 1e-4:
        want_its, want_res = _sor_cycles_restated(oracle, x, idx, p["mask"], lut, 1, 1e-4, 200000, halve=True)
    assert (cycles, its, res) == (want_cycles, want_its, np.float32(want_res)) and res <= 1e-4
    assert cycles >= 3 and its > 0            # the rule needs two earlier residuals; at these sizes the sweeps always finish
    assert_bit_equal(down(d), x, "auto")


def test_randomised_geometries_all_extension_methods(ctx, oracle, lut):
    """40 random draws (shape incl. 1-pixel-wide and tiny, Dirichlet density from almost none to almost all, random start,
    contraction, method): V-cycles with residual checks (and so the extrapolation), SOR with a random factor, SOR cycles and
    the automatic method -- every result, count and residual identical to the restatement's."""
    from test_gpu_parity import _sor_cycles_restated
    rng = np.random.default_rng(20261004)
    for trial in range(40):
        rows = int(rng.integers(1, 330)); cols = int(rng.integers(1, 400))
        if trial % 9 == 0: rows = int(rng.integers(1, 5))
        if trial % 13 == 0: cols = int(rng.integers(1, 5))
        contract = int(rng.integers(0, 2))
        p = make_problem(rows, cols, seed=3000 + trial, coverage=float(rng.choice([0.002, 0.05, 0.1, 0.5, 0.95])))
        if (p["mask"] == 255).sum() == 0:
            p["mask"][rows // 2, cols // 2] = 255; p["depth"][rows // 2, cols // 2] = 128
        free = p["mask"] != 255
        p["depth"][free] = rng.uniform(0, 255, int(free.sum())).astype(np.float32)
        ctx.GPUAllocateDeviceMemory(rows, cols, 1)
        ctx.set_option(rt.OPT_FP_CONTRACT, contract)
        idx = oracle.index_to_weight(p["gray"], None, 0, 0)
        d, m, g = up(p["depth"]), up(p["mask"]), up(p["gray"])
        x = p["depth"].copy()
        kind = trial % 4
        what = f"trial {trial}: {rows}x{cols} contract {contract} kind {kind}"
        if kind == 0:                                   # V-cycles to a residual (check every cycle -> extrapolation allowed)
            its, res = ctx.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_MULTIGRID, maxIterations=12, tolerance=1e-4)
            want_its, want_res, _ = oracle.mg_solve(x, idx, p["mask"], lut, contract, 12, 1e-4, 1)
            assert (its, res) == (want_its, np.float32(want_res)), what
        elif kind == 1:                                 # red-black SOR, fixed count
            om = float(rng.uniform(0.5, 1.98)); n = int(rng.integers(1, 40))
            its, _ = ctx.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=n, relaxation=om)
            for _ in range(n):
                oracle.rbgs_sweep(x, idx, p["mask"], lut, contract, om)
            assert its == n, what
        elif kind == 2:                                 # SOR cycles, capped
            its, res = ctx.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=1500, tolerance=1e-4, relaxation=rt.RELAXATION_AUTO)
            want_its, want_res = _sor_cycles_restated(oracle, x, idx, p["mask"], lut, contract, 1e-4, 1500)
            assert (its, res) == (want_its, np.float32(want_res)), what
        else:                                           # automatic: V-cycles, then SOR cycles of half length, capped
            its, res = ctx.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_AUTO, maxIterations=1200, tolerance=1e-4)
            sor_seconds, cycle_seconds = ctx.auto_model(rows, cols)
            want_cycles, want_res, _ = oracle.mg_solve(x, idx, p["mask"], lut, contract, 60, 1e-4, 1, alternative_seconds=sor_seconds, cycle_seconds=cycle_seconds)
            want_its = 0
            if want_res > 1e-4:
                want_its, want_res = _sor_cycles_restated(oracle, x, idx, p["mask"], lut, contract, 1e-4, 1200, halve=True)
            assert (ctx.last_cycles, its, res) == (want_cycles, want_its, np.float32(want_res)), what
        assert_bit_equal(down(d), x, what)
    ctx.set_option(rt.OPT_FP_CONTRACT, 1)
