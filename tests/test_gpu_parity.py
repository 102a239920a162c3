"""GPU parity tests proper (-m gpu): every call goes through the C ABI of librtdd.so and is
compared with the CPU oracle on the same seeded inputs.

Stated tolerance (BASELINE.json north_star): depth maps within 1e-4 max-abs.  Because the HIP
kernels repeat the oracle's f32 operations one for one, the tests assert the stronger property
-- BIT-EXACT equality -- for the solver, the index pass, the annotation kernels and the three
depth effects (haze since round 3: one deterministic exp on both sides)."""
import ctypes as C

import numpy as np
import pytest

import realtimedepthdiffusion_amd as rt
from gpu_util import assert_bit_equal, down, up
from realtimedepthdiffusion_amd.synth import make_problem

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def _ctx():
    c = rt.Context(0)
    c.GPULoadWeights(0.4)
    yield c
    c.close()


@pytest.fixture
def ctx(_ctx):
    """The shared context with every option back at its default: no test may depend on what the one before it left set."""
    _ctx.set_option(rt.OPT_FP_CONTRACT, 1); _ctx.set_option(rt.OPT_PERSISTENT, 1)
    for k in (rt.OPT_SWEEP_KERNEL, rt.OPT_TILE, rt.OPT_TEMPORAL_DEPTH, rt.OPT_ROWS_PER_WAVE):
        _ctx.set_option(k, 0)
    _ctx.profile_enable(False)
    return _ctx


def _solve_gpu(ctx, p, iters, level, levels, contract, align=512, opts=None):
    rows, cols = p["gray"].shape
    ctx.GPUAllocateDeviceMemory(rows << level, cols << level, levels)
    ctx.set_option(rt.OPT_FP_CONTRACT, contract)
    for k, v in (opts or {}).items():
        ctx.set_option(k, v)
    d, m, g = up(p["depth"], align), up(p["mask"], align), up(p["gray"], align)
    ctx.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, iters, 1e-5, level)
    ctx.synchronize()
    return down(d)


@pytest.mark.parametrize("contract", [1, 0])
@pytest.mark.parametrize("shape,level,levels,iters", [
    ((64, 64), 2, 3, 100),      # coarsest level: un-gated weights
    ((33, 47), 1, 3, 61),       # odd sizes, gated rule with threshold 4, odd iteration count
    ((40, 40), 0, 3, 40),       # level 0: gated rule with threshold 0
    ((1, 9), 0, 1, 12), ((9, 1), 0, 1, 12), ((1, 1), 0, 1, 3), ((2, 2), 0, 1, 5),
    ((67, 120), 4, 5, 1000),    # the 1080p cascade's coarsest level, full 1000 sweeps
    ((130, 257), 0, 1, 50),     # crosses the 256-pixel strip boundary by one pixel
    ((270, 480), 0, 1, 250),
    ((5, 1030), 0, 1, 30),
])
def test_solver_bit_exact(ctx, oracle, lut, shape, level, levels, iters, contract):
    p = make_problem(shape[0], shape[1], seed=100 + shape[0] + shape[1])
    if (p["mask"] == 255).sum() == 0:
        p["mask"][0, 0] = 255; p["depth"][0, 0] = 64
    if level != levels - 1:                       # give the depth gate something to bite on
        rng = np.random.default_rng(1)
        free = p["mask"] != 255
        p["depth"][free] = rng.uniform(0, 255, free.sum()).astype(np.float32)
    want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], iters, level, levels - 1, lut, contract, threads=min(8, oracle.max_threads()))
    got = _solve_gpu(ctx, p, iters, level, levels, contract)
    assert np.abs(got - want).max() <= TOL
    assert_bit_equal(got, want, f"solver {shape} level {level}")


@pytest.mark.parametrize("tile,depth", [(1, 1), (1, 4), (1, 7), (1, 8), (1, 16), (2, 4), (2, 8), (2, 5), (3, 4), (3, 12), (3, 16), (12, 8), (13, 8), (13, 5), (9, 28), (9, 24), (11, 8), (4, 28), (14, 8), (14, 28), (14, 5), (14, 1), (15, 8), (15, 12), (15, 1), (16, 8), (16, 5), (16, 20),
                                        (5, 8), (5, 12), (6, 8), (6, 12), (7, 8), (7, 12), (8, 8), (8, 12), (10, 8), (10, 12), (4, 8), (4, 12)])
@pytest.mark.parametrize("shape,iters,level,levels", [((200, 333), 37, 0, 1), ((67, 120), 64, 1, 2), ((300, 130), 24, 0, 2), ((129, 129), 19, 0, 1)])
def test_blocked_kernel_bit_exact(ctx, oracle, lut, tile, depth, shape, iters, level, levels):
    """Temporal blocking is only a re-schedule: any tile shape / depth must reproduce the oracle bit for bit."""
    p = make_problem(shape[0], shape[1], seed=shape[0] * 7 + shape[1])
    if level != levels - 1:
        rng = np.random.default_rng(2)
        free = p["mask"] != 255
        p["depth"][free] = rng.uniform(0, 255, free.sum()).astype(np.float32)
    want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], iters, level, levels - 1, lut, 1, threads=min(8, oracle.max_threads()))
    got = _solve_gpu(ctx, p, iters, level, levels, 1, opts={rt.OPT_SWEEP_KERNEL: 2, rt.OPT_TILE: tile, rt.OPT_TEMPORAL_DEPTH: depth})
    for k in (rt.OPT_SWEEP_KERNEL, rt.OPT_TILE, rt.OPT_TEMPORAL_DEPTH):
        ctx.set_option(k, 0)
    assert_bit_equal(got, want, f"blocked tile {tile} depth {depth} {shape}")


@pytest.mark.parametrize("shape,iters,tile,depth", [
    ((200, 333), 100, 9, 8),       # 7x5 tiles of 64x64, 12.5 blocks
    ((200, 333), 37, 1, 4),        # odd tail block
    ((270, 480), 250, 4, 8),       # 5x4 tiles of 128x96
    ((135, 240), 500, 9, 16),      # depth 16: 8x5 tiles of 32x32 centres
    ((540, 960), 125, 4, 8),
    ((1080, 1920), 200, 4, 8),     # 252 workgroups: the headline configuration
    ((1080, 1920), 120, 12, 8),    # the same tile held by 12 waves of 16 pixels per thread
    ((1080, 1920), 120, 13, 8),    # ... and by 8 waves of 24 pixels per thread
    ((270, 480), 50, 13, 8),
    ((300, 130), 64, 7, 16),
])
def test_persistent_mode_bit_exact(ctx, oracle, lut, shape, iters, tile, depth):
    """Persistent mode: one launch, tiles stay in registers, neighbouring workgroups trade halo strips through
    memory every `depth` sweeps (agent-scope flags).  Still only a re-schedule: bit-exact."""
    p = make_problem(shape[0], shape[1], seed=shape[0] + 3 * shape[1])
    want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], iters, 0, 0, lut, 1, threads=oracle.max_threads())
    opts = {rt.OPT_SWEEP_KERNEL: 2, rt.OPT_TILE: tile, rt.OPT_TEMPORAL_DEPTH: depth, rt.OPT_PERSISTENT: 1}
    ctx.profile_enable(True)
    for rep in range(3):                       # repeated: flags are re-zeroed per call, L1/L2 are warm on later reps
        got = _solve_gpu(ctx, p, iters, 0, 1, 1, opts=opts)
        assert ctx.profile().launches == 1, "not persistent: more than one launch"
        assert_bit_equal(got, want, f"persistent {shape} tile {tile} depth {depth} rep {rep}")
    ctx.profile_enable(False)
    for k in opts:
        ctx.set_option(k, 0)


def test_randomised_shapes_and_options(ctx, oracle, lut):
    """60 random (shape, sweeps, level rule, kernel, tile, depth, persistence, contraction) draws, fixed seed: every
    one bit-exact.  Catches geometry corner cases the hand-picked lists miss (ragged tiles, tiny centres, 1-pixel strips)."""
    rng = np.random.default_rng(20261003)
    for trial in range(60):
        rows = int(rng.integers(1, 420)); cols = int(rng.integers(1, 520))
        if trial % 7 == 0: rows = int(rng.integers(1, 6))
        if trial % 11 == 0: cols = int(rng.integers(1, 6))
        iters = int(rng.integers(1, 90))
        levels = int(rng.integers(1, 4)); level = int(rng.integers(0, levels))
        contract = int(rng.integers(0, 2))
        kernel = int(rng.choice([0, 0, 1, 2]))
        opts = {rt.OPT_SWEEP_KERNEL: kernel, rt.OPT_PERSISTENT: int(rng.integers(0, 2))}
        if kernel != 1 and rng.random() < 0.6:
            opts[rt.OPT_TILE] = int(rng.integers(1, 14)); opts[rt.OPT_TEMPORAL_DEPTH] = int(rng.choice([1, 2, 3, 4, 6, 8, 12, 16, 24]))
        p = make_problem(rows, cols, seed=1000 + trial)
        if (p["mask"] == 255).sum() == 0:
            p["mask"][rows // 2, cols // 2] = 255; p["depth"][rows // 2, cols // 2] = 128
        free = p["mask"] != 255
        p["depth"][free] = rng.uniform(0, 255, int(free.sum())).astype(np.float32)
        want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], iters, level, levels - 1, lut, contract, threads=min(8, oracle.max_threads()))
        got = _solve_gpu(ctx, p, iters, level, levels, contract, opts=opts)
        for k in (rt.OPT_SWEEP_KERNEL, rt.OPT_TILE, rt.OPT_TEMPORAL_DEPTH):
            ctx.set_option(k, 0)
        ctx.set_option(rt.OPT_PERSISTENT, 1)
        assert_bit_equal(got, want, f"trial {trial}: {rows}x{cols} iters {iters} level {level}/{levels} contract {contract} opts {opts}")


def test_randomised_column_kernel(ctx, oracle, lut):
    """The column-layout kernel (tile 14) under 40 random (shape, sweeps, depth, level rule, contraction) draws: single tiles
    (<= 64 x 64, every wave stays to the end), ragged multi-tile grids, depths from 1 to 28 (waves leave as the needed region shrinks)."""
    rng = np.random.default_rng(20261004)
    for trial in range(40):
        rows = int(rng.integers(1, 300)); cols = int(rng.integers(1, 300))
        if trial % 4 == 0: rows, cols = int(rng.integers(1, 65)), int(rng.integers(1, 65))
        iters = int(rng.integers(1, 70))
        levels = int(rng.integers(1, 4)); level = int(rng.integers(0, levels))
        contract = int(rng.integers(0, 2))
        opts = {rt.OPT_SWEEP_KERNEL: 2, rt.OPT_TILE: 14, rt.OPT_TEMPORAL_DEPTH: int(rng.choice([1, 2, 3, 5, 8, 12, 16, 24, 28]))}
        p = make_problem(rows, cols, seed=2000 + trial)
        if (p["mask"] == 255).sum() == 0:
            p["mask"][rows // 2, cols // 2] = 255; p["depth"][rows // 2, cols // 2] = 128
        free = p["mask"] != 255
        p["depth"][free] = rng.uniform(0, 255, int(free.sum())).astype(np.float32)
        want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], iters, level, levels - 1, lut, contract, threads=min(8, oracle.max_threads()))
        got = _solve_gpu(ctx, p, iters, level, levels, contract, opts=opts)
        info = ctx.last_solve_info()
        for k in (rt.OPT_SWEEP_KERNEL, rt.OPT_TILE, rt.OPT_TEMPORAL_DEPTH):
            ctx.set_option(k, 0)
        assert info.tile == 14, info.describe()
        assert_bit_equal(got, want, f"trial {trial}: {rows}x{cols} iters {iters} level {level}/{levels} contract {contract} opts {opts}")


def test_persistent_mode_stress_under_uneven_load(oracle, lut):
    """Hand-offs must hold under uneven load with warm caches (cdna_hip_programming.md G16 pitfall 3): 40 solves
    back to back (1000 halo exchanges of 252 workgroups) while a second stream streams 1 GiB through the memory
    system, every result compared bit for bit."""
    import torch
    rows, cols, iters = 1080, 1920, 200
    p = make_problem(rows, cols, seed=99)
    want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], iters, 0, 0, lut, 1, threads=oracle.max_threads())
    c = rt.Context(0); c.GPULoadWeights(0.4); c.GPUAllocateDeviceMemory(rows, cols, 1)
    main = torch.cuda.Stream(); side = torch.cuda.Stream()
    c.set_stream(main.cuda_stream)
    m, g = up(p["mask"]), up(p["gray"])
    ds = [up(p["depth"]) for _ in range(40)]
    noise = torch.empty(256 << 20, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    c.profile_enable(True)
    for i, d in enumerate(ds):
        if i % 3 == 0:
            with torch.cuda.stream(side):
                noise.mul_(1.0001)                      # bandwidth hog on another stream, overlapping some solves
        c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, iters, 0.0, 0)
        assert c.profile().launches == 1                # persistent
    c.synchronize(); torch.cuda.synchronize()           # synchronize() also checks the kernel's timeout word
    bad = [i for i, d in enumerate(ds) if not np.array_equal(down(d).view(np.uint32), want.view(np.uint32))]
    assert not bad, f"solves {bad} differ"
    c.close()


@pytest.mark.parametrize("kernel", [1, 2])
@pytest.mark.parametrize("contract", [0, 1])
def test_both_kernels_both_contractions(ctx, oracle, lut, kernel, contract):
    p = make_problem(150, 260, seed=8)
    want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 45, 0, 0, lut, contract, threads=min(8, oracle.max_threads()))
    got = _solve_gpu(ctx, p, 45, 0, 1, contract, opts={rt.OPT_SWEEP_KERNEL: kernel})
    ctx.set_option(rt.OPT_SWEEP_KERNEL, 0)
    assert_bit_equal(got, want)


def test_residual_stop_and_rbgs_extensions(ctx, oracle, lut):
    """Extensions (no reference behaviour): residual-stopped Jacobi and red-black Gauss-Seidel."""
    p = make_problem(96, 128, seed=12)
    p["gray"] = np.ascontiguousarray(p["gray"] >> 4)             # well-conditioned (see tests/test_oracle.py)
    rows, cols = 96, 128
    ctx.GPUAllocateDeviceMemory(rows, cols, 1)
    ctx.set_option(rt.OPT_FP_CONTRACT, 1)
    idx = oracle.index_to_weight(p["gray"], None, 0, 0)
    # Chebyshev-Jacobi with a residual stop: stops early, and the reported residual is the oracle's
    d, m, g = up(p["depth"]), up(p["mask"]), up(p["gray"])
    its, res = ctx.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_CHEBYSHEV_JACOBI, maxIterations=4000, tolerance=2e-3, checkEvery=20)
    got = down(d)
    assert its < 4000 and its % 20 == 0 and res <= 2e-3
    assert res == np.float32(oracle.residual(got, idx, p["mask"], lut, 1))
    want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], its, 0, 0, lut, 1)
    assert_bit_equal(got, want, "residual-stopped solve == fixed-count solve of the same length")
    # red-black Gauss-Seidel: bit-exact against the oracle's sweep
    d = up(p["depth"])
    its, _ = ctx.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=25, tolerance=0.0)
    x = p["depth"].copy()
    for _ in range(25):
        oracle.rbgs_sweep(x, idx, p["mask"], lut, 1)
    assert its == 25
    assert_bit_equal(down(d), x, "rbgs")
    # and to a residual
    d = up(p["depth"])
    its, res = ctx.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=20000, tolerance=1e-4, checkEvery=50)
    assert res <= 1e-4 and its < 20000
    assert oracle.residual(down(d), idx, p["mask"], lut, 1) <= 1e-4


@pytest.mark.parametrize("rows,cols,sweeps", [(64, 128, 7), (128, 128, 9), (129, 128, 5), (96, 131, 6), (240, 333, 11),
                                             (547, 1021, 13), (1080, 1920, 26), (50, 77, 3), (1, 300, 4), (300, 1, 4), (2, 2, 5)])
@pytest.mark.parametrize("contract,omega", [(1, 1.0), (0, 1.0), (1, 1.7), (0, 1.93)])
def test_rbgs_blocked_bit_exact(ctx, oracle, lut, rows, cols, sweeps, contract, omega):
    """Register-blocked red-black Gauss-Seidel (csrc/rbgs_blocked.hip; an EXTENSION, north_star config 3) == the
    oracle's in-place sweep, bit for bit: single-tile and multi-tile shapes, sweep counts that are not multiples of the
    4 sweeps one launch carries, ragged edges, and the one-launch-per-colour fallback kernel beside it."""
    p = make_problem(rows, cols, seed=rows * 7 + cols)
    ctx.GPUAllocateDeviceMemory(rows, cols, 1)
    ctx.set_option(rt.OPT_FP_CONTRACT, contract)
    idx = oracle.index_to_weight(p["gray"], None, 0, 0)
    x = p["depth"].copy()
    for _ in range(sweeps):
        oracle.rbgs_sweep(x, idx, p["mask"], lut, contract, omega)      # omega != 1: the SOR step
    m, g = up(p["mask"]), up(p["gray"])
    # (sweep kernel, tile shape, sweeps per launch): automatic; one launch per colour; both tile shapes at several depths
    for kernel, tile, depth in ((0, 0, 0), (1, 0, 0), (0, 1, 4), (0, 1, 3), (0, 2, 8), (0, 2, 5), (0, 1, 12), (0, 2, 24)):
        ctx.set_option(rt.OPT_SWEEP_KERNEL, kernel); ctx.set_option(rt.OPT_TILE, tile); ctx.set_option(rt.OPT_TEMPORAL_DEPTH, depth)
        d = up(p["depth"])
        its, _ = ctx.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=sweeps, tolerance=0.0, relaxation=omega)
        assert its == sweeps
        assert_bit_equal(down(d), x, f"rbgs kernel {kernel} tile {tile} depth {depth} {rows}x{cols}x{sweeps} omega {omega}")
    ctx.set_option(rt.OPT_SWEEP_KERNEL, 0); ctx.set_option(rt.OPT_TILE, 0); ctx.set_option(rt.OPT_TEMPORAL_DEPTH, 0)
    ctx.set_option(rt.OPT_FP_CONTRACT, 1)


@pytest.mark.parametrize("rows,cols,tile,depth", [(1080, 1920, 0, 0), (1080, 1920, 2, 6), (720, 1280, 1, 4), (900, 1600, 2, 4)])
def test_rbgs_persistent_mode_bit_exact(ctx, oracle, lut, rows, cols, tile, depth):
    """The red-black kernel's persistent mode (one launch, halo strips traded every `depth` sweeps) against the oracle's sweep,
    with the launch-per-block path beside it; sweep counts that end in a short last block; plain and over-relaxed."""
    p = make_problem(rows, cols, seed=rows + cols)
    ctx.GPUAllocateDeviceMemory(rows, cols, 1)
    idx = oracle.index_to_weight(p["gray"], None, 0, 0)
    m, g = up(p["mask"]), up(p["gray"])
    ctx.set_option(rt.OPT_TILE, tile); ctx.set_option(rt.OPT_TEMPORAL_DEPTH, depth)
    for sweeps, omega in ((27, 1.0), (32, 1.9)):
        x = p["depth"].copy()
        for _ in range(sweeps):
            oracle.rbgs_sweep(x, idx, p["mask"], lut, 1, omega)
        for persistent in (1, 0):
            ctx.set_option(rt.OPT_PERSISTENT, persistent)
            ctx.profile_enable(True)
            for rep in range(2):
                d = up(p["depth"])
                ctx.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=sweeps, tolerance=0.0, relaxation=omega)
                launches = ctx.profile().launches
                assert (launches == 1) == bool(persistent), (launches, persistent)
                assert_bit_equal(down(d), x, f"rbgs persistent={persistent} {rows}x{cols} tile {tile} depth {depth} x{sweeps} omega {omega} rep {rep}")
            ctx.profile_enable(False)
    ctx.set_option(rt.OPT_PERSISTENT, 1); ctx.set_option(rt.OPT_TILE, 0); ctx.set_option(rt.OPT_TEMPORAL_DEPTH, 0)


def _sor_cycles_restated(oracle, x, idx, mask, lut, contract, tol, max_its, halve=False):
    """rtdd_solve_ex's RTDD_RELAXATION_AUTO schedule (include/rtdd.h, csrc/api.cpp) driven through the oracle's sweep and residual."""
    import math
    rows, cols = x.shape
    longest = max(rows, cols)
    w0 = min(1.99, max(1.0, 2.0 / (1.0 + math.sin(4.0 * math.pi / longest))))
    done, res = 0, float("nan")

    def run(n, om):
        nonlocal done
        n = min(n, max_its - done)
        for _ in range(n):
            oracle.rbgs_sweep(x, idx, mask, lut, contract, float(om))
        done += n
    reached, cycle = False, 0
    while done < max_its and not reached:
        e = min(cycle, 6)
        gap = max(0.005, (2.0 - w0) / (1 << e))
        w_hi, w_mid = np.float32(2.0 - gap), max(np.float32(1.0), np.float32(2.0 - 10.0 * gap))
        base = (longest + 1) // 2 if halve else longest             # RTDD_METHOD_AUTO starts its SOR cycles at half length
        run(base << e, w_hi); run((base << e) // 4, w_mid)
        for _ in range(5):
            if done >= max_its or reached:
                break
            run(20, 1.0)
            res = oracle.residual(x, idx, mask, lut, contract)
            reached = res <= tol
        cycle += 1
    return done, res


@pytest.mark.parametrize("rows,cols,seed", [(96, 128, 12), (200, 150, 3)])
def test_sor_cycles_reach_the_residual_and_match_the_restated_schedule(ctx, oracle, lut, rows, cols, seed):
    """BASELINE config 3 in the small: red-black SOR cycles + Gauss-Seidel polish to a 1e-4 residual (an EXTENSION).  Same
    sweep count, same reported residual and bit-identical depth as the schedule restated over the oracle; the oracle's
    residual of the GPU result is under the tolerance; plain Gauss-Seidel needs far more sweeps for the same residual."""
    p = make_problem(rows, cols, seed=seed)
    ctx.GPUAllocateDeviceMemory(rows, cols, 1)
    ctx.set_option(rt.OPT_FP_CONTRACT, 1)
    idx = oracle.index_to_weight(p["gray"], None, 0, 0)
    d, m, g = up(p["depth"]), up(p["mask"]), up(p["gray"])
    its, res = ctx.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=20000, tolerance=1e-4, relaxation=rt.RELAXATION_AUTO)
    got = down(d)
    x = p["depth"].copy()
    want_its, want_res = _sor_cycles_restated(oracle, x, idx, p["mask"], lut, 1, 1e-4, 20000)
    assert (its, res) == (want_its, np.float32(want_res)) and res <= 1e-4
    assert_bit_equal(got, x, "SOR cycles")
    assert oracle.residual(got, idx, p["mask"], lut, 1) <= 1e-4
    d = up(p["depth"])
    its_gs, res_gs = ctx.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=its, tolerance=1e-4, checkEvery=its)
    assert res_gs > 1e-4, "plain Gauss-Seidel should not get there in the same number of sweeps"


def test_solver_zero_iterations_returns_input(ctx):
    p = make_problem(32, 48, seed=5)
    got = _solve_gpu(ctx, p, 0, 0, 1, 1)
    assert_bit_equal(got, p["depth"])


@pytest.mark.parametrize("align", [4, 64, 512, 4096])
def test_solver_pitch_independent(ctx, oracle, lut, align):
    p = make_problem(50, 77, seed=9)
    want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 35, 0, 0, lut, 1)
    rows, cols = 50, 77
    ctx.GPUAllocateDeviceMemory(rows, cols, 1)
    ctx.set_option(rt.OPT_FP_CONTRACT, 1)
    # u8 images with pitch == cols (align 1) exercise unaligned rows; depth pitch stays a multiple of 4
    d, m, g = up(p["depth"], align), up(p["mask"], 1 if align == 4 else align), up(p["gray"], 1 if align == 4 else align)
    ctx.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 35, 0.0, 0)
    assert_bit_equal(down(d), want)


@pytest.mark.parametrize("rows_per_wave", [1, 3, 7, 16, 64])
def test_solver_independent_of_strip_height(ctx, oracle, lut, rows_per_wave):
    p = make_problem(75, 300, seed=17)
    want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 20, 0, 0, lut, 1)
    got = _solve_gpu(ctx, p, 20, 0, 1, 1, opts={rt.OPT_SWEEP_KERNEL: 1, rt.OPT_ROWS_PER_WAVE: rows_per_wave})   # the one-sweep kernel
    ctx.set_option(rt.OPT_ROWS_PER_WAVE, 0); ctx.set_option(rt.OPT_SWEEP_KERNEL, 0)
    assert_bit_equal(got, want)


def test_solver_smaller_problem_in_larger_allocation(ctx, oracle, lut):
    """GPUMatrixFreeSolver may be called with rows/cols below the level's allocation."""
    p = make_problem(37, 61, seed=4)
    want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 25, 1, 2, lut, 1)
    ctx.GPUAllocateDeviceMemory(200, 300, 3)
    ctx.set_option(rt.OPT_FP_CONTRACT, 1)
    d, m, g = up(p["depth"]), up(p["mask"]), up(p["gray"])
    ctx.GPUMatrixFreeSolver(d, m, g, 37, 61, 0.4, 25, 0.0, 1)
    assert_bit_equal(down(d), want)


def test_solver_denormal_weights_survive(ctx, oracle, lut):
    """|gray difference| >= 219 selects f32-denormal weights; flushing them would change the mean."""
    rows, cols = 16, 64
    gray = np.zeros((rows, cols), np.uint8); gray[:, ::2] = 255          # every horizontal edge is 255
    gray[::2, :] //= 1
    mask = np.full((rows, cols), 32, np.uint8); mask[:, 0] = 255; mask[:, -1] = 255
    depth = np.full((rows, cols), 255, np.float32); depth[:, 0] = 0; depth[:, -1] = 192
    p = {"gray": gray, "mask": mask, "depth": depth}
    want = oracle.solve(depth.copy(), mask, gray, 50, 0, 0, lut, 1)
    got = _solve_gpu(ctx, p, 50, 0, 1, 1)
    assert_bit_equal(got, want)


@pytest.mark.parametrize("kernel", [1, 2])
def test_denormal_divisor_takes_the_full_divide(ctx, oracle, lut, kernel):
    """Checkerboard gray: all four weights of every pixel are the denormal w[255] = 5.6e-45, so the
    divisor sum(w) is itself denormal and the blocked kernel must leave its hoisted-reciprocal path."""
    rows, cols = 40, 72
    yy, xx = np.mgrid[0:rows, 0:cols]
    gray = (((yy + xx) & 1) * 255).astype(np.uint8)
    mask = np.full((rows, cols), 32, np.uint8); mask[::9, ::7] = 255
    rng = np.random.default_rng(6)
    depth = rng.uniform(0, 255, (rows, cols)).astype(np.float32)
    p = {"gray": gray, "mask": mask, "depth": depth}
    assert lut[255] * 4 < 1.2e-38                         # really denormal
    want = oracle.solve(depth.copy(), mask, gray, 40, 0, 0, lut, 1)
    got = _solve_gpu(ctx, p, 40, 0, 1, 1, opts={rt.OPT_SWEEP_KERNEL: kernel})
    ctx.set_option(rt.OPT_SWEEP_KERNEL, 0)
    assert_bit_equal(got, want, "denormal divisor")


@pytest.mark.parametrize("shape", [(24, 24), (60, 200)])
def test_values_decaying_through_the_denormal_range(ctx, oracle, lut, shape):
    """All labels 0: the iterate decays towards 0 through 1e-31 .. denormals .. exact zero, which
    drives the weighted sums below 2^-103 where the divide needs hardware rescaling."""
    rows, cols = shape
    rng = np.random.default_rng(3)
    gray = rng.integers(100, 104, (rows, cols), dtype=np.uint8)
    mask = np.full((rows, cols), 32, np.uint8); mask[0, :] = mask[-1, :] = 255; mask[:, 0] = mask[:, -1] = 255; mask[::5, ::5] = 255
    depth = np.where(mask == 255, 0.0, 255.0).astype(np.float32)
    p = {"gray": gray, "mask": mask, "depth": depth}
    iters = 1500
    want = oracle.solve(depth.copy(), mask, gray, iters, 0, 0, lut, 1)
    tiny = np.abs(want[mask != 255])
    assert tiny.max() < 1e-30 and (tiny > 0).any()        # the test really is in the tiny regime
    got = _solve_gpu(ctx, p, iters, 0, 1, 1)
    assert_bit_equal(got, want, "decay to denormals")


@pytest.mark.parametrize("level,levels", [(0, 1), (0, 3), (1, 3), (2, 3)])
def test_index_to_weight_bit_exact(ctx, oracle, level, levels):
    import torch
    p = make_problem(45, 83, seed=23)
    rng = np.random.default_rng(3)
    depth = rng.uniform(-20, 280, p["gray"].shape).astype(np.float32)     # includes out-of-range values: saturating cast
    depth[::4, ::3] = np.floor(depth[::4, ::3])
    want = oracle.index_to_weight(p["gray"], depth, level, levels - 1)
    ctx.GPUAllocateDeviceMemory(45 << level, 83 << level, levels)
    idx = torch.zeros((45, 83, 2), dtype=torch.int32, device="cuda:0")
    ctx.index_to_weight(up(p["gray"]), up(depth), idx, level, 45, 83)
    ctx.synchronize()
    assert np.array_equal(idx.cpu().numpy(), want)


def test_full_size_1080p_1000_sweeps_matches_oracle(ctx, oracle, lut):
    """BASELINE config 2 at full size: 1920x1080, one level, 1000 sweeps, against the oracle on all host cores."""
    p = make_problem(1080, 1920, seed=1234)
    want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 1000, 0, 0, lut, 1, threads=oracle.max_threads())
    got = _solve_gpu(ctx, p, 1000, 0, 1, 1)
    assert np.abs(got - want).max() <= TOL
    assert_bit_equal(got, want, "1080p x 1000")
    # size-independent properties: Dirichlet pixels untouched, result within the label hull (+ Chebyshev overshoot slack)
    dir_ = p["mask"] == 255
    assert np.array_equal(got[dir_], p["depth"][dir_])
    assert got.min() > -64 and got.max() < 320


def _fresh(rows, cols, withhold):
    c = rt.Context(0)
    c.GPUAllocateDeviceMemory(rows, cols, 1); c.GPULoadWeights(0.4)
    if withhold:
        c.set_option(rt.OPT_DEBUG_POLL_LIMIT_US, 3000); c.set_option(rt.OPT_DEBUG_WITHHOLD_TILE, 100 + 1)
    return c


def test_persistent_timeout_heals_itself(oracle, lut, capfd):
    """The reference's GPUMatrixFreeSolver always leaves a valid depth map (src/GPUSolver.cu:311-314).  With one tile's flag withheld
    (RTDD_OPT_DEBUG_WITHHOLD_TILE) its neighbours run into the poll limit: the launch and the work queued behind it drain at once,
    the copy-back kernels store nothing, and the next synchronising call runs the affected solves again without persistence and
    returns RTDD_OK with the oracle's bits -- one warning, persistence off for the context from then on."""
    import time
    rows, cols = 1080, 1920
    p = make_problem(rows, cols, seed=5)
    m, g = up(p["mask"]), up(p["gray"])
    with _fresh(rows, cols, False) as c:                     # no knob: persistent, no heal
        d = up(p["depth"])
        c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 64, 0.0, 0); c.synchronize()
        info = c.last_solve_info()
        assert info.kernel == 2 and info.persistent == 1 and info.iterations == 64, info.describe()
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 0 and c.get_option(rt.OPT_PERSISTENT) == 1
    capfd.readouterr()
    with _fresh(rows, cols, True) as c:
        d2 = up(p["depth"])
        t0 = time.perf_counter()
        for _ in range(3):                                   # the failed solve and two more queued behind it, each on the one before's result
            c.GPUMatrixFreeSolver(d2, m, g, rows, cols, 0.4, 400, 0.0, 0)
        c.synchronize()                                      # heals: no error
        elapsed = time.perf_counter() - t0
        assert elapsed < 0.5, f"a timed-out launch must drain quickly and its replay is three short solves, took {elapsed:.2f} s"
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 1 and c.get_option(rt.OPT_PERSISTENT) == 0
        want = p["depth"].copy()
        for _ in range(3):
            want = oracle.solve(want, p["mask"], p["gray"], 400, 0, 0, lut, 1, threads=oracle.max_threads())
        assert_bit_equal(down(d2), want, "three queued solves behind a timed-out launch")
        err = capfd.readouterr().err
        assert err.count("rtdd: persistent sweep kernel") == 1, err
        # from now on: one launch per block of sweeps, no further heal, no further warning
        d3 = up(p["depth"])
        c.GPUMatrixFreeSolver(d3, m, g, rows, cols, 0.4, 64, 0.0, 0); c.synchronize()
        info = c.last_solve_info()
        assert info.kernel == 2 and info.persistent == 0, info.describe()
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 1 and "rtdd:" not in capfd.readouterr().err
        assert_bit_equal(down(d3), oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 64, 0, 0, lut, 1, threads=oracle.max_threads()), "solve after a heal")


def test_a_launch_that_is_not_resident_is_found_out_at_its_start(oracle, lut):
    """Round 6: a persistent workgroup's first act is to announce itself and to see its eight neighbours announce themselves, with a bound
    of 1.5 ms (persist_sync.hpp kArrivalPollLimit), before any sweep -- on a shared GPU a launch that is not fully resident costs that, not
    the 200 ms an exchange may wait.  With a tile's flag withheld and NO debug poll limit set (the default bounds are in force): the solve's
    launch, its heal and the replay without persistence together take a few milliseconds."""
    import time
    rows, cols = 1080, 1920
    p = make_problem(rows, cols, seed=6)
    m, g = up(p["mask"]), up(p["gray"])
    with rt.Context(0) as c:
        c.GPUAllocateDeviceMemory(rows, cols, 1); c.GPULoadWeights(0.4)
        d = up(p["depth"])
        c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 8, 0.0, 0); c.synchronize()      # (first-call costs out of the way)
        c.set_option(rt.OPT_DEBUG_WITHHOLD_TILE, 57 + 1)
        d = up(p["depth"])
        t0 = time.perf_counter()
        c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 200, 0.0, 0)
        c.synchronize()
        elapsed = time.perf_counter() - t0
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 1
        assert elapsed < 0.05, f"{elapsed * 1e3:.1f} ms: the arrival bound is 1.5 ms, the replay of 200 sweeps ~0.5 ms (the old bound alone was 200 ms)"
        assert_bit_equal(down(d), oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 200, 0, 0, lut, 1, threads=oracle.max_threads()), "healed solve")


def test_persistent_timeout_leaves_the_input_until_the_heal(oracle, lut):
    """Between the failed launch and the synchronising call the caller's buffer holds the solve's INPUT (k_finish stores nothing),
    which is what lets the solve run again; rtdd_download heals too and then copies again."""
    import torch
    rows, cols = 1080, 1920
    p = make_problem(rows, cols, seed=8)
    m, g = up(p["mask"]), up(p["gray"])
    with _fresh(rows, cols, True) as c:
        d = up(p["depth"])
        c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 100, 0.0, 0)
        torch.cuda.synchronize()                             # (not a library call: the status word has not been looked at)
        assert_bit_equal(down(d), p["depth"], "a timed-out solve must not touch the caller's depth")
        host = np.empty((rows, cols), np.float32)
        c._check(rt.lib().rtdd_download(c._h, C.c_void_p(host.ctypes.data), C.c_size_t(cols * 4), C.c_void_p(d.data_ptr()), C.c_size_t(d.stride(0) * 4), C.c_size_t(cols * 4), C.c_int(rows)))
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 1
        assert_bit_equal(host, oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 100, 0, 0, lut, 1, threads=oracle.max_threads()), "rtdd_download after a heal")


@pytest.mark.parametrize("method", ["jacobi_residual", "red_black"])
def test_persistent_timeout_heals_inside_a_residual_stopped_solve_and_in_red_black(method):
    """A residual-stopped solve synchronises itself: the time-out is found by its own first residual check and the solve starts over;
    the red-black kernel's persistent mode shares the hand-off.  Compared with the same solve on a context that never was persistent."""
    rows, cols = 1080, 1920
    p = make_problem(rows, cols, seed=5)
    m, g = up(p["mask"]), up(p["gray"])

    def run(c):
        d = up(p["depth"])
        if method == "jacobi_residual":
            out = c.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_CHEBYSHEV_JACOBI, maxIterations=400, tolerance=1e-3, checkEvery=200)
        else:
            out = c.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=64, tolerance=0.0)
        c.synchronize()
        return out, down(d)

    with _fresh(rows, cols, False) as c:
        c.set_option(rt.OPT_PERSISTENT, 0)
        want_out, want = run(c)
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 0
    with _fresh(rows, cols, True) as c:
        got_out, got = run(c)
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 1 and c.get_option(rt.OPT_PERSISTENT) == 0
    assert got_out[0] == want_out[0] and (got_out[1] == want_out[1] or (np.isnan(got_out[1]) and np.isnan(want_out[1])))
    assert_bit_equal(got, want, method)


def test_status_one_behind_a_launch_per_block_solve_heals_too(oracle, lut):
    """Status 1 forced behind a NON-persistent launch (a small level queued behind a failed persistent one looks like this): the
    solve is run again all the same and the result is the oracle's."""
    rows, cols = 270, 480
    p = make_problem(rows, cols, seed=6)
    with _fresh(rows, cols, False) as c:
        d, m, g = up(p["depth"]), up(p["mask"]), up(p["gray"])
        c.set_option(rt.OPT_DEBUG_FORCE_STATUS, 1)
        c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 24, 0.0, 0)
        c.synchronize()                                      # healed: the forced word is one shot
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 1
        assert_bit_equal(down(d), oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 24, 0, 0, lut, 1, threads=oracle.max_threads()), "forced status 1")


@pytest.mark.parametrize("rows,cols,persistent", [(2160, 3840, 0), (270, 480, 0), (1080, 1920, 1)])
def test_status_word_is_read_after_every_blocked_launch(ctx, oracle, lut, rows, cols, persistent):
    """The intra-workgroup wait of the blocked kernel is bounded in the launch-per-block instantiation too (status 2).  Whatever
    the mode, the next synchronising call must report it -- not a later, unrelated persistent launch (round-2 advisor finding)."""
    p = make_problem(rows, cols, seed=6)
    ctx.GPUAllocateDeviceMemory(rows, cols, 1)
    ctx.set_option(rt.OPT_PERSISTENT, persistent)
    d, m, g = up(p["depth"]), up(p["mask"]), up(p["gray"])
    ctx.set_option(rt.OPT_DEBUG_FORCE_STATUS, 2)
    ctx.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 24, 0.0, 0)
    info = ctx.last_solve_info()
    assert info.kernel == 2 and info.persistent == persistent, info.describe()
    with pytest.raises(rt.RtddError) as e:
        ctx.synchronize()
    assert e.value.status == rt.RTDD_ERR_TIMEOUT and "neighbouring wave" in str(e.value)
    ctx.synchronize()                                                  # reported once, then cleared
    want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 24, 0, 0, lut, 1, threads=oracle.max_threads())
    d2 = up(p["depth"])
    ctx.GPUMatrixFreeSolver(d2, m, g, rows, cols, 0.4, 24, 0.0, 0); ctx.synchronize()
    assert_bit_equal(down(d2), want, "solve after a reported status")


def test_solve_info_names_the_path_that_ran(ctx):
    """rtdd_solve_info / rtdd_last_solve_info: kernel, tile, depth, persistence and contraction of the solve that just ran."""
    p = make_problem(270, 480, seed=2)
    rows, cols = 270, 480
    ctx.GPUAllocateDeviceMemory(rows, cols, 1)
    d, m, g = up(p["depth"]), up(p["mask"]), up(p["gray"])
    ctx.set_option(rt.OPT_SWEEP_KERNEL, 1)
    ctx.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 10, 0.0, 0)
    i = ctx.last_solve_info()
    assert (i.kernel, i.iterations, i.launches, i.persistent, i.fp_contract) == (1, 10, 10, 0, 1), i.describe()
    ctx.set_option(rt.OPT_SWEEP_KERNEL, 2); ctx.set_option(rt.OPT_TILE, 9); ctx.set_option(rt.OPT_TEMPORAL_DEPTH, 4); ctx.set_option(rt.OPT_PERSISTENT, 0)
    ctx.set_option(rt.OPT_FP_CONTRACT, 0)
    ctx.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 10, 0.0, 0)
    i = ctx.last_solve_info()
    assert (i.kernel, i.tile, i.iterations, i.launches, i.persistent, i.fp_contract) == (2, 9, 10, 3, 0, 0), i.describe()
    ctx.set_option(rt.OPT_SWEEP_KERNEL, 0); ctx.set_option(rt.OPT_TILE, 0); ctx.set_option(rt.OPT_TEMPORAL_DEPTH, 0); ctx.set_option(rt.OPT_PERSISTENT, 1)
    ctx.set_option(rt.OPT_FP_CONTRACT, 1)
    its, _ = ctx.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=12, tolerance=0.0)
    i = ctx.last_solve_info()
    assert (i.kernel, i.iterations, i.fp_contract) == (4, 12, 1), i.describe()
    ctx.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,iters,small", [(67, 120, 56, True), (135, 240, 48, True), (1080, 1920, 16, False), (2160, 3840, 8, False)])
def test_automatic_configuration_keeps_the_column_kernel_to_the_small_levels(ctx, rows, cols, iters, small):
    """The cost model offers the column-layout kernel (tile 14) only where every tile of a launch is resident at once: the two
    coarsest levels of a 1080p cascade.  At 1080p and 4K it is far slower than the large tiles (a round-2 regression caught by the
    profile run: 4K 1100 -> 794 Gpx-it/s) -- the choice is part of the contract."""
    p = make_problem(rows, cols, seed=3)
    ctx.GPUAllocateDeviceMemory(rows, cols, 1)
    d, m, g = up(p["depth"]), up(p["mask"]), up(p["gray"])
    ctx.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, iters, 0.0, 0)
    i = ctx.last_solve_info()
    ctx.synchronize()
    assert i.kernel == 2, i.describe()
    assert (i.tile == 14) == small, i.describe()


def test_acquire_variant_of_the_hand_off_is_bit_exact(oracle, lut, tmp_path):
    """The persistent kernels' hand-off has no agent-scope acquire (16-byte sc1 loads only; validated on gfx950 in SPX mode -- DESIGN.md).
    Its documented fallback, -DRTDD_EXCHANGE_ACQUIRE=1 (one acquire by wave 0 + plain loads, in k_sweep_blocked AND k_rbgs_blocked), is
    built as librtdd_acq.so by __graft_entry__.build() and must produce the same bits: a persistent 1080p Jacobi solve against the
    oracle and a persistent red-black solve against the default library, in a process of its own (RTDD_LIBRARY selects the library)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "realtimedepthdiffusion_amd", "librtdd_acq.so")
    if not os.path.exists(so):
        pytest.skip("librtdd_acq.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    rows, cols = 1080, 1920
    p = make_problem(rows, cols, seed=12)
    want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 200, 0, 0, lut, 1, threads=oracle.max_threads())
    with _fresh(rows, cols, False) as c:
        d = up(p["depth"])
        c.solve_ex(d, up(p["mask"]), up(p["gray"]), rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=64, tolerance=0.0); c.synchronize()
        assert c.last_solve_info().persistent == 1
        rb_want = down(d)
    np.save(tmp_path / "want.npy", want); np.save(tmp_path / "rb_want.npy", rb_want)
    code = f"""
import sys, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem
from gpu_util import up, down, assert_bit_equal
p = make_problem({rows}, {cols}, seed=12)
with rt.Context(0) as c:
    c.GPUAllocateDeviceMemory({rows}, {cols}, 1); c.GPULoadWeights(0.4)
    m, g = up(p["mask"]), up(p["gray"])
    d = up(p["depth"]); c.GPUMatrixFreeSolver(d, m, g, {rows}, {cols}, 0.4, 200, 0.0, 0); c.synchronize()
    assert c.last_solve_info().persistent == 1
    assert_bit_equal(down(d), np.load({str(tmp_path / 'want.npy')!r}), "Jacobi, acquire variant")
    d = up(p["depth"]); c.solve_ex(d, m, g, {rows}, {cols}, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=64, tolerance=0.0); c.synchronize()
    assert c.last_solve_info().persistent == 1
    assert_bit_equal(down(d), np.load({str(tmp_path / 'rb_want.npy')!r}), "red-black, acquire variant")
print("acquire variant ok")
"""
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RTDD_LIBRARY=so), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "acquire variant ok" in out.stdout, out.stdout + out.stderr


def test_a_logged_solve_is_healed_before_its_weight_table_changes(oracle, lut):
    """Calls that change what a logged (not yet confirmed) solve ran on settle the log first: a timed-out solve followed by GPULoadWeights
    with another beta must be replayed with the table it was made with -- the result is the oracle's for beta = 0.4, and the next solve
    uses the new table."""
    rows, cols = 1080, 1920
    p = make_problem(rows, cols, seed=14)
    m, g = up(p["mask"]), up(p["gray"])
    with _fresh(rows, cols, True) as c:
        d = up(p["depth"])
        c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 80, 0.0, 0)
        c.GPULoadWeights(0.2)                                            # settles the log: the solve above is healed here
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 1
        assert_bit_equal(down(d), oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 80, 0, 0, lut, 1, threads=oracle.max_threads()), "healed with the table of its call")
        d2 = up(p["depth"])
        c.GPUMatrixFreeSolver(d2, m, g, rows, cols, 0.2, 80, 0.0, 0); c.synchronize()
        assert_bit_equal(down(d2), oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 80, 0, 0, oracle.load_weights(0.2), 1, threads=oracle.max_threads()), "the next solve: new table")


def test_effects_queued_behind_a_timed_out_solve_are_replayed_with_it(oracle, lut):
    """solve -> defocus / desaturation / haze of its depth map -> synchronise, all asynchronous, and the solve's persistent launch times
    out: the effects ran on the solve's INPUT.  They are logged behind the unconfirmed solve and run again behind its replay: after the
    synchronising call all three images are the oracle's effects of the oracle's depth map."""
    rows, cols = 1080, 1920
    p = make_problem(rows, cols, seed=17)
    rgb = np.random.default_rng(5).integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 60, 0, 0, lut, 1, threads=oracle.max_threads())
    with _fresh(rows, cols, True) as c:
        d, m, g, o = up(p["depth"]), up(p["mask"]), up(p["gray"]), up(rgb)
        arts = [up(np.zeros_like(rgb)) for _ in range(3)]
        c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 60, 0.0, 0)
        c.GPUSimulateDefocus(o, d, arts[0], rows, cols)
        c.GPUSimulateDesaturation(o, g, d, arts[1], rows, cols)
        c.GPUSimulateHaze(o, d, arts[2], rows, cols)
        c.synchronize()
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 1
        assert_bit_equal(down(d), want, "the healed solve")
        assert np.array_equal(down(arts[0]), oracle.defocus(rgb, want, threads=oracle.max_threads())), "defocus behind a healed solve"
        assert np.array_equal(down(arts[1]), oracle.desaturate(rgb, p["gray"], want, 1)), "desaturation behind a healed solve"
        assert np.array_equal(down(arts[2]), oracle.haze(rgb, want, 1)), "haze behind a healed solve"


def test_a_second_time_out_during_the_replay_is_reported(oracle, lut):
    """RTDD_OPT_DEBUG_FORCE_STATUS = 3: status 1 behind the solve and again behind its replay.  One heal is attempted, the second failure
    is final: RTDD_ERR_TIMEOUT, the caller's depth untouched (the replay's copy-back stored nothing either), and the context works
    afterwards -- without persistence."""
    rows, cols = 270, 480
    p = make_problem(rows, cols, seed=19)
    with _fresh(rows, cols, False) as c:
        d, m, g = up(p["depth"]), up(p["mask"]), up(p["gray"])
        c.set_option(rt.OPT_DEBUG_FORCE_STATUS, 3)
        c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 40, 0.0, 0)
        with pytest.raises(rt.RtddError) as e:
            c.synchronize()
        assert e.value.status == rt.RTDD_ERR_TIMEOUT and "again" in str(e.value), e.value
        assert_bit_equal(down(d), p["depth"], "a solve that failed twice must leave its input")
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 1 and c.get_option(rt.OPT_PERSISTENT) == 0
        c.synchronize()                                                  # reported once
        c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 40, 0.0, 0); c.synchronize()
        assert_bit_equal(down(d), oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 40, 0, 0, lut, 1, threads=4), "the context works afterwards")


def test_persistence_is_rearmed_after_a_heal_with_a_bounded_back_off(oracle, lut, capfd):
    """One time-out must not cost an unchanged main.cpp the persistent kernel until exit: after a heal persistent
    launches are suspended for RTDD_OPT_PERSISTENT_REARM_AFTER solves, then tried again; a second time-out doubles the wait, the
    fifth switches persistence off for good.  Every result on the way is the oracle's."""
    rows, cols = 1080, 1920
    p = make_problem(rows, cols, seed=23)
    m, g = up(p["mask"]), up(p["gray"])
    want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 24, 0, 0, lut, 1, threads=oracle.max_threads())

    def solve(c):
        d = up(p["depth"])
        c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 24, 0.0, 0); c.synchronize()
        assert_bit_equal(down(d), want, "solve on the way")
        return c.last_solve_info().persistent

    with _fresh(rows, cols, True) as c:
        c.set_option(rt.OPT_PERSISTENT_REARM_AFTER, 2)
        assert solve(c) == 0 and c.get_option(rt.OPT_TIMEOUT_HEALS) == 1          # timed out, healed (the replay is not persistent)
        assert c.get_option(rt.OPT_PERSISTENT) == 0 and c.get_option(rt.OPT_PERSISTENT_SUSPENDED) == 2
        c.set_option(rt.OPT_DEBUG_WITHHOLD_TILE, 0)                              # the GPU is no longer "shared"
        assert solve(c) == 0 and c.get_option(rt.OPT_PERSISTENT_SUSPENDED) == 1
        assert solve(c) == 0 and c.get_option(rt.OPT_PERSISTENT_SUSPENDED) == 0 and c.get_option(rt.OPT_PERSISTENT) == 1   # re-armed for the next one
        assert solve(c) == 1 and c.get_option(rt.OPT_TIMEOUT_HEALS) == 1
        # shared again: the back-off doubles (4, 8, 16), the fifth time-out is the last
        c.set_option(rt.OPT_DEBUG_WITHHOLD_TILE, 100 + 1)
        for heals, wait in ((2, 4), (3, 8), (4, 16)):
            assert solve(c) == 0 and c.get_option(rt.OPT_TIMEOUT_HEALS) == heals and c.get_option(rt.OPT_PERSISTENT_SUSPENDED) == wait
            for _ in range(wait):
                assert solve(c) == 0
            assert c.get_option(rt.OPT_PERSISTENT) == 1
        assert solve(c) == 0 and c.get_option(rt.OPT_TIMEOUT_HEALS) == 5 and c.get_option(rt.OPT_PERSISTENT_SUSPENDED) == -1
        for _ in range(3):
            assert solve(c) == 0
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 5 and c.get_option(rt.OPT_PERSISTENT) == 0
        c.set_option(rt.OPT_DEBUG_WITHHOLD_TILE, 0); c.set_option(rt.OPT_PERSISTENT, 1)      # said explicitly: armed at once
        assert solve(c) == 1 and c.get_option(rt.OPT_PERSISTENT_SUSPENDED) == 0
    assert capfd.readouterr().err.count("rtdd: persistent sweep kernel") == 1                # one warning per context


def test_with_healing_switched_off_a_time_out_is_reported_and_nothing_is_remembered(oracle, lut):
    """RTDD_OPT_TIMEOUT_HEAL = 0 (a host that cannot keep its buffers alive until an rtdd call has synchronised): no call is
    logged, the time-out comes back as RTDD_ERR_TIMEOUT, the depth keeps the solve's input, and the context works afterwards."""
    rows, cols = 1080, 1920
    p = make_problem(rows, cols, seed=29)
    m, g = up(p["mask"]), up(p["gray"])
    with _fresh(rows, cols, True) as c:
        c.set_option(rt.OPT_TIMEOUT_HEAL, 0)
        d = up(p["depth"])
        c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 40, 0.0, 0)
        assert c.get_option(rt.OPT_PENDING_CALLS) == 0
        with pytest.raises(rt.RtddError) as e:
            c.synchronize()
        assert e.value.status == rt.RTDD_ERR_TIMEOUT and "RTDD_OPT_TIMEOUT_HEAL" in str(e.value), e.value
        assert_bit_equal(down(d), p["depth"], "a reported time-out leaves the input")
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 1 and c.get_option(rt.OPT_PERSISTENT) == 0
        c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 40, 0.0, 0); c.synchronize()
        assert_bit_equal(down(d), oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 40, 0, 0, lut, 1, threads=oracle.max_threads()), "the context works afterwards")


def test_confirmed_calls_leave_the_log_without_a_library_synchronisation(oracle, lut):
    """The log of calls a heal would run again holds calls IN FLIGHT, not history (it kept the caller's pointers for up to 4096
    calls until an rtdd-level synchronisation): the copy-back kernel of a solve reports its sequence number in page-locked memory, and
    the next logging call drops everything up to it -- after a torch / hipDeviceSynchronize of the caller's own, too."""
    import torch
    rows, cols = 270, 480
    p = make_problem(rows, cols, seed=31)
    with _fresh(rows, cols, False) as c:
        d, m, g = up(p["depth"]), up(p["mask"]), up(p["gray"])
        rgb = up(np.zeros((rows, cols, 3), np.uint8)); art = up(np.zeros((rows, cols, 3), np.uint8))
        for _ in range(40):
            c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 8, 0.0, 0)
            c.GPUSimulateHaze(rgb, d, art, rows, cols)
        torch.cuda.synchronize()                              # not a library call
        assert c.get_option(rt.OPT_PENDING_CALLS) <= 1        # (the haze behind the last solve: it leaves with the next confirmed solve)
        c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 8, 0.0, 0)
        torch.cuda.synchronize()
        assert c.get_option(rt.OPT_PENDING_CALLS) == 0
        c.synchronize()
        want = p["depth"].copy()
        for _ in range(41):
            want = oracle.solve(want, p["mask"], p["gray"], 8, 0, 0, lut, 1, threads=4)
        assert_bit_equal(down(d), want, "41 chained solves")


def test_set_stream_settles_the_log_on_the_old_stream(oracle, lut):
    """rtdd_ctx_set_stream with an unconfirmed, timed-out solve queued on the old stream: the heal runs there, before anything is
    queued on the new stream."""
    import torch
    rows, cols = 1080, 1920
    p = make_problem(rows, cols, seed=37)
    m, g = up(p["mask"]), up(p["gray"])
    with _fresh(rows, cols, True) as c:
        d = up(p["depth"])
        c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 40, 0.0, 0)
        s2 = torch.cuda.Stream()
        c.set_stream(s2.cuda_stream)
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 1 and c.get_option(rt.OPT_PENDING_CALLS) == 0
        want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 40, 0, 0, lut, 1, threads=oracle.max_threads())
        assert_bit_equal(down(d), want, "healed on the old stream")
        c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 40, 0.0, 0); c.synchronize()
        assert_bit_equal(down(d), oracle.solve(want, p["mask"], p["gray"], 40, 0, 0, lut, 1, threads=oracle.max_threads()), "next solve on the new stream")
        c.set_stream(0)

