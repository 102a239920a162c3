"""GPU parity for the annotation kernels and the depth effects (-m gpu), through the C ABI."""
import numpy as np
import pytest

import realtimedepthdiffusion_amd as rt
from gpu_util import down, up
from realtimedepthdiffusion_amd.synth import make_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = rt.Context(0)
    yield c
    c.close()


def _rgb(rows, cols, seed):
    return np.random.default_rng(seed).integers(0, 256, (rows, cols, 3), dtype=np.uint8)


@pytest.mark.parametrize("shape", [(9, 11), (64, 64), (67, 120), (135, 240)])
def test_convert_to_float(ctx, oracle, shape):
    rows, cols = shape
    edited = _rgb(rows, cols, 1)
    mask = np.where(np.random.default_rng(2).random(shape) < 0.2, 255, 32).astype(np.uint8)
    dst = np.random.default_rng(3).uniform(0, 255, shape).astype(np.float32)
    want = oracle.convert_to_float(edited, dst.copy(), mask)
    d = up(dst)
    ctx.GPUConvertToFloat(up(edited), d, up(mask), rows, cols)
    assert np.array_equal(down(d), want)


@pytest.mark.parametrize("fine", [(8, 8), (9, 11), (135, 240), (624, 672), (853, 1280)])
def test_pyrdown_annotation(ctx, oracle, fine):
    rows, cols = fine
    crows, ccols = rows // 2, cols // 2
    rng = np.random.default_rng(rows)
    edited = _rgb(rows, cols, 5)
    mask = np.where(rng.random(fine) < 0.15, 255, 32).astype(np.uint8)
    cm = np.zeros((crows, ccols), np.uint8); ce = np.zeros((crows, ccols, 3), np.uint8)
    cm[0, 0] = 255; ce[0, 0] = 9                        # stale state must survive (never cleared)
    wm, we = cm.copy(), ce.copy()
    oracle.pyrdown_annotation(mask, edited, wm, we)
    gm, ge = up(cm), up(ce)
    ctx.GPUPyrDownAnnotation(up(mask), up(edited), rows, cols, gm, ge, crows, ccols)
    assert np.array_equal(down(gm), wm) and np.array_equal(down(ge), we)


@pytest.mark.parametrize("x,y,r", [(50, 40, 21), (0, 0, 9), (99, 79, 10), (-5, 30, 20), (300, 300, 8), (10, 10, 0), (10, 10, 1), (20, 20, -6)])
def test_paint_image(ctx, oracle, x, y, r):
    rows, cols = 80, 100
    e = _rgb(rows, cols, 7); m = np.full((rows, cols), 32, np.uint8)
    we, wm = e.copy(), m.copy()
    oracle.paint_image(x, y, 192, r, we, wm)
    ge, gm = up(e), up(m)
    ctx.GPUPaintImage(x, y, 192, r, ge, gm, rows, cols)
    assert np.array_equal(down(ge), we) and np.array_equal(down(gm), wm)


def _depth(rows, cols, seed):
    p = make_problem(rows, cols, seed=seed)
    rng = np.random.default_rng(seed)
    d = (p["depth"] * rng.uniform(0.0, 1.0, (rows, cols))).astype(np.float32)
    d[::7, ::5] = 255.0; d[1::7, ::5] = 0.0
    return d, p["gray"]


@pytest.mark.parametrize("contract", [1, 0])
@pytest.mark.parametrize("shape", [(6, 8), (67, 120), (270, 480), (700, 560)])
def test_desaturation_bit_exact(ctx, oracle, shape, contract):
    rows, cols = shape
    depth, gray = _depth(rows, cols, 31)
    orig = _rgb(rows, cols, 11)
    ctx.set_option(rt.OPT_FP_CONTRACT, contract)
    art = up(np.zeros_like(orig))
    ctx.GPUSimulateDesaturation(up(orig), up(gray), up(depth), art, rows, cols)
    assert np.array_equal(down(art), oracle.desaturate(orig, gray, depth, contract))


@pytest.mark.parametrize("contract", [1, 0])
@pytest.mark.parametrize("shape", [(6, 8), (67, 120), (270, 480), (700, 560)])
def test_haze_bit_exact(ctx, oracle, shape, contract):
    """Bit-exact since round 3: the transmission exp(-2 d / 255) is one fixed sequence of f64 operations on both sides."""
    rows, cols = shape
    depth, _ = _depth(rows, cols, 37)
    depth[::5, ::3] = np.random.default_rng(3).uniform(-400, 700, depth[::5, ::3].shape).astype(np.float32)    # out-of-range depths: t > 1 and t -> 0
    orig = _rgb(rows, cols, 13)
    ctx.set_option(rt.OPT_FP_CONTRACT, contract)
    art = up(np.zeros_like(orig))
    ctx.GPUSimulateHaze(up(orig), up(depth), art, rows, cols)
    assert np.array_equal(down(art), oracle.haze(orig, depth, contract))


def test_haze_transmission_every_depth_pattern(ctx, oracle):
    """Every f32 depth in steps over [-64, 320] plus the special values, against white and black pixels: with o = 0 the output byte is
    trunc((1 - t) * 255), with o = 255 it is trunc(t * 255 + (1 - t) * 255): 4.2 M depths x 2 colours, all equal to the oracle's."""
    bits = np.arange(0, 1 << 22, dtype=np.uint32)
    d = (np.float32(-64.0) + bits.astype(np.float64) * (384.0 / (1 << 22))).astype(np.float32)
    d[:8] = [0.0, -0.0, 255.0, 1e-30, np.float32(np.inf), np.float32(-np.inf), np.float32(np.nan), 3e38]
    rows, cols = 2048, 2048
    depth = d.reshape(rows, cols)
    orig = np.zeros((rows, cols, 3), np.uint8); orig[:, :, 1] = 255; orig[:, :, 2] = 128
    art = up(np.zeros_like(orig))
    ctx.set_option(rt.OPT_FP_CONTRACT, 1)
    ctx.GPUSimulateHaze(up(orig), up(depth), art, rows, cols)
    assert np.array_equal(down(art), oracle.haze(orig, depth, 1))


# RTDD_OPT_DEFOCUS_PATH: 1 = the global summed-area table (four launches), 2 = per-tile tables in LDS (one launch; what images up to
# ~1080p take by default).  Every defocus test runs both.
DEFOCUS_PATHS = [1, 2]


def _defocus(ctx, path, orig_dev, depth_dev, art_dev, rows, cols):
    ctx.set_option(rt.OPT_DEFOCUS_PATH, path)
    try:
        ctx.GPUSimulateDefocus(orig_dev, depth_dev, art_dev, rows, cols)
        ctx.synchronize()
    finally:
        ctx.set_option(rt.OPT_DEFOCUS_PATH, 0)


@pytest.mark.parametrize("path", DEFOCUS_PATHS)
@pytest.mark.parametrize("shape", [(6, 8), (80, 80), (67, 120), (270, 480), (700, 560)])
def test_defocus_bit_exact(ctx, oracle, shape, path):
    rows, cols = shape
    depth, _ = _depth(rows, cols, 41)
    orig = _rgb(rows, cols, 17)
    art = up(np.zeros_like(orig))
    _defocus(ctx, path, up(orig), up(depth), art, rows, cols)
    assert np.array_equal(down(art), oracle.defocus(orig, depth, threads=min(8, oracle.max_threads())))


@pytest.mark.parametrize("path", DEFOCUS_PATHS)
@pytest.mark.parametrize("shape,align", [((67, 121), 1), ((131, 259), 1), ((40, 4099), 4), ((9, 70), 1), ((300, 66), 2), ((17, 64), 1), ((16, 128), 4), ((33, 63), 4)])
def test_defocus_unaligned_rows_and_ragged_tiles(ctx, oracle, shape, align, path):
    """Caller pitches that are no multiple of 4 bytes (the byte paths of the table build and of the lookup), widths that leave a
    ragged last 64-pixel tile and a ragged last group of four, and a row wider than one sweep of the build's workgroup (4096 px:
    the per-row carry between column ranges; too large a window for the tile kernel, which hands that one to the table)."""
    rows, cols = shape
    depth, _ = _depth(rows, cols, 43)
    orig = _rgb(rows, cols, 18)
    art = up(np.zeros_like(orig), align)
    _defocus(ctx, path, up(orig, align), up(depth), art, rows, cols)
    assert np.array_equal(down(art), oracle.defocus(orig, depth, threads=min(8, oracle.max_threads())))


@pytest.mark.parametrize("path", DEFOCUS_PATHS)
def test_defocus_windows_beyond_one_lookup(ctx, oracle, path):
    """Windows of more than 8224 pixels (the packed table's 21-bit fields) are summed in strips: a small image with a depth far
    above 255 makes every window the whole (clipped) image -- 160 x 120 = 19 200 pixels, three strips.  In the tile kernel every one
    of these pixels is beyond its region: the wave sums the window from the image."""
    rows, cols = 120, 160
    orig = _rgb(rows, cols, 20)
    depth = np.full((rows, cols), 4000.0, np.float32); depth[::7, ::5] = 900.0; depth[3::11, 1::3] = 255.0
    art = up(np.zeros_like(orig))
    _defocus(ctx, path, up(orig), up(depth), art, rows, cols)
    assert np.array_equal(down(art), oracle.defocus(orig, depth, threads=min(8, oracle.max_threads())))


@pytest.mark.parametrize("path", DEFOCUS_PATHS)
def test_defocus_out_of_range_depth_is_defined(ctx, oracle, path):
    rows, cols = 120, 160
    orig = _rgb(rows, cols, 19)
    depth = np.random.default_rng(5).uniform(-300, 600, (rows, cols)).astype(np.float32)
    art = up(np.zeros_like(orig))
    _defocus(ctx, path, up(orig), up(depth), art, rows, cols)
    assert np.array_equal(down(art), oracle.defocus(orig, depth, threads=min(8, oracle.max_threads())))


def test_defocus_automatic_path_survives_a_depth_that_is_no_depth(oracle):
    """Automatic path, a fresh context: the tile kernel meets windows beyond its region (summed by hand, flagged), the context hears of
    it at its next synchronisation and uses the table from then on -- same bits before and after, also for a sane depth map."""
    rows, cols = 96, 140
    orig = _rgb(rows, cols, 31)
    crazy = np.full((rows, cols), 3.0e5, np.float32); crazy[::3, ::2] = 2000.0
    sane, _ = _depth(rows, cols, 59)
    want = {id(d): oracle.defocus(orig, d, threads=min(8, oracle.max_threads())) for d in (crazy, sane)}
    c = rt.Context(0)
    try:
        assert c.get_option(rt.OPT_DEFOCUS_LAST_PATH) == 0
        # which kernel the automatic choice launched is observable (RTDD_OPT_DEFOCUS_LAST_PATH): the tile kernel until a synchronisation
        # has seen its "window beyond the region" flag, the table from then on, the tile kernel again once the option is set to 0 again
        for depth, path in ((sane, 2), (crazy, 2), (crazy, 1), (sane, 1)):
            art = up(np.zeros_like(orig))
            c.GPUSimulateDefocus(up(orig), up(depth), art, rows, cols)
            assert c.get_option(rt.OPT_DEFOCUS_LAST_PATH) == path, (path, c.get_option(rt.OPT_DEFOCUS_LAST_PATH))
            c.synchronize()
            assert np.array_equal(down(art), want[id(depth)])
        c.set_option(rt.OPT_DEFOCUS_PATH, 0)
        art = up(np.zeros_like(orig))
        c.GPUSimulateDefocus(up(orig), up(sane), art, rows, cols); c.synchronize()
        assert c.get_option(rt.OPT_DEFOCUS_LAST_PATH) == 2 and np.array_equal(down(art), want[id(sane)])
    finally:
        c.close()


@pytest.mark.parametrize("shape", [(1080, 1920), (853, 1280), (1440, 1754)])
def test_defocus_tile_kernel_equals_table_at_size(ctx, oracle, shape):
    """The largest sizes the tile kernel takes (1440 x 1754: kernelSize 56, half-width 28 = its limit): every pixel equal to the table
    path's, a depth map with a patch of out-of-range values (windows beyond the region) and special values, and 400 sampled pixels
    against the oracle's literal gather."""
    rows, cols = shape
    rng = np.random.default_rng(rows)
    depth, _ = _depth(rows, cols, 47)
    depth[100:140, 200:260] = rng.uniform(255, 700, (40, 60)).astype(np.float32)
    depth[0, :8] = [0.0, -0.0, 255.0, 1e-30, np.float32(np.inf), np.float32(-np.inf), np.float32(np.nan), 3e38]
    orig = _rgb(rows, cols, 21)
    o, dd = up(orig), up(depth)
    a1, a2 = up(np.zeros_like(orig)), up(np.zeros_like(orig))
    _defocus(ctx, 1, o, dd, a1, rows, cols)
    _defocus(ctx, 2, o, dd, a2, rows, cols)
    got = down(a2)
    assert np.array_equal(got, down(a1))
    ys = rng.integers(0, rows, 400); xs = rng.integers(0, cols, 400)
    ys[:60] = rng.integers(95, 145, 60); xs[:60] = rng.integers(195, 265, 60)
    keep = ~((ys == 0) & (xs < 8))       # (the infinite depths' windows are the whole image: sums beyond 2^24, where the reference's own f32 accumulation rounds)
    ys, xs = ys[keep], xs[keep]
    want = oracle.defocus_at(orig, depth, ys, xs)
    assert np.array_equal(got[ys, xs], want)


def test_effects_1080p_properties(ctx):
    """Full-size, size-independent properties: depth 0 is the identity for all three effects."""
    rows, cols = 1080, 1920
    orig = _rgb(rows, cols, 23)
    gray = np.random.default_rng(29).integers(0, 256, (rows, cols), dtype=np.uint8)
    zero = up(np.zeros((rows, cols), np.float32))
    o = up(orig)
    for call in ("haze", "desat", "defocus"):
        art = up(np.zeros_like(orig))
        if call == "haze":
            ctx.GPUSimulateHaze(o, zero, art, rows, cols)
        elif call == "desat":
            ctx.GPUSimulateDesaturation(o, up(gray), zero, art, rows, cols)
        else:
            ctx.GPUSimulateDefocus(o, zero, art, rows, cols)
        assert np.array_equal(down(art), orig), call
    # depth 255 desaturation is the gray image
    art = up(np.zeros_like(orig))
    ctx.GPUSimulateDesaturation(o, up(gray), up(np.full((rows, cols), 255, np.float32)), art, rows, cols)
    assert np.array_equal(down(art), np.repeat(gray[..., None], 3, 2))


def test_error_behaviour(ctx):
    c = rt.Context(0)
    d = up(np.zeros((8, 8), np.float32)); m = up(np.zeros((8, 8), np.uint8))
    with pytest.raises(rt.RtddError) as e:
        c.GPUMatrixFreeSolver(d, m, m, 8, 8, 0.4, 10, 0.0, 0)            # before allocate
    assert e.value.status == 2
    c.GPUAllocateDeviceMemory(8, 8, 1)
    with pytest.raises(rt.RtddError) as e:
        c.GPUMatrixFreeSolver(d, m, m, 8, 8, 0.4, 10, 0.0, 0)            # before load_weights
    assert e.value.status == 2
    c.GPULoadWeights(0.4)
    with pytest.raises(rt.RtddError) as e:
        c.GPUMatrixFreeSolver(d, m, m, 8, 8, 0.4, 10, 0.0, 3)            # level out of range
    assert e.value.status == 1
    with pytest.raises(rt.RtddError):
        c.GPUMatrixFreeSolver(d, m, m, 64, 64, 0.4, 10, 0.0, 0)          # larger than the allocation
    c.GPUMatrixFreeSolver(d, m, m, 8, 8, 0.4, 10, 0.0, 0)                # and a legal call still works
    c.synchronize()
    # rtdd_solve_ex (extension) argument checks: unknown method, relaxation outside [0,2), AUTO without a tolerance
    for kw in (dict(method=7), dict(method=rt.METHOD_RED_BLACK_GS, relaxation=2.0), dict(method=rt.METHOD_RED_BLACK_GS, relaxation=-0.5),
               dict(method=rt.METHOD_AUTO, tolerance=0.0), dict(maxIterations=-1)):
        with pytest.raises(rt.RtddError) as e:
            c.solve_ex(d, m, m, 8, 8, 0, **kw)
        assert e.value.status == 1, kw
    with pytest.raises(rt.RtddError):
        c.multigrid_level(0, 0)                                          # no multigrid solve has run on this context
    assert c.solve_ex(d, m, m, 8, 8, 0, method=rt.METHOD_MULTIGRID, maxIterations=2)[0] == 2
    assert c.multigrid_level(0, 4).shape == (8, 8)
    with pytest.raises(rt.RtddError):
        c.multigrid_level(5, 0)
    c.close()


def test_dropin_cxx_symbols_run(oracle, lut):
    """Call the reference's own mangled symbols (what main.cpp links against) through ctypes."""
    import ctypes as C
    L = rt.lib()
    p = make_problem(48, 64, seed=77)
    want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 30, 0, 0, lut, 1)
    d, m, g = up(p["depth"]), up(p["mask"]), up(p["gray"])
    vp, sz, i32, f32 = C.c_void_p, C.c_size_t, C.c_int, C.c_float
    L._Z23GPUAllocateDeviceMemoryiii.restype = None
    L._Z23GPUAllocateDeviceMemoryiii(i32(48), i32(64), i32(1))
    L._Z14GPULoadWeightsf.restype = None
    L._Z14GPULoadWeightsf(f32(0.4))
    f = L._Z19GPUMatrixFreeSolverPfmPhmS0_miififi
    f.restype = None
    f(vp(d.data_ptr()), sz(d.stride(0) * 4), vp(m.data_ptr()), sz(m.stride(0)), vp(g.data_ptr()), sz(g.stride(0)),
      i32(48), i32(64), f32(0.4), i32(30), f32(1e-5), i32(0))
    assert np.array_equal(down(d).view(np.uint32), want.view(np.uint32))      # returns synchronised, like the reference
    L._Z19GPUFreeDeviceMemoryi.restype = None
    L._Z19GPUFreeDeviceMemoryi(i32(1))


def test_defocus_refuses_a_table_it_cannot_address():
    """The whole-image defocus table (8 bytes per pixel) is addressed with 32-bit byte offsets: an image whose table would reach 4 GiB
    (about 536 million pixels; rtdd_simulate_defocus's own size check admits rows x cols up to 2^30) is refused with RTDD_ERR_INVALID
    before anything is allocated or launched -- not answered with wrapped offsets and zeros.  (Only with RTDD_OPT_DEFOCUS_SLICE_MB = 0:
    by default such an image is built and looked up slice by slice, and every slice's table is small.  The pointers below are never
    dereferenced BECAUSE the call is refused: nothing of this kind may be tried with the slices on.)"""
    import ctypes as C
    import torch
    rows = cols = 30000                               # rows^2 + cols^2 = 1.8e9 < 2^31: passes the effect's size check; table = 7.2 GB
    small = torch.zeros(1024, dtype=torch.uint8, device="cuda:0")
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        c.set_option(rt.OPT_DEFOCUS_SLICE_MB, 0)
        assert c.get_option(rt.OPT_DEFOCUS_SLICE_MB) == 0
        L = rt.lib()
        args = (C.c_void_p(small.data_ptr()), C.c_size_t(cols * 3), C.c_void_p(small.data_ptr()), C.c_size_t(cols * 4),
                C.c_void_p(small.data_ptr() + 512), C.c_size_t(cols * 3), C.c_int(rows), C.c_int(cols))
        assert L.rtdd_simulate_defocus(c._h, *args) == 1          # RTDD_ERR_INVALID
        assert b"too large for the defocus table" in L.rtdd_last_error(c._h)
        c.synchronize()


def test_defocus_windows_wider_than_a_strip():
    """A window of more than 8224 pixels is summed in strips that share corner rows (the packed 3 x 21-bit difference holds 8224 x 255);
    one that is WIDER than 8224 pixels -- an image wider than that and a depth far above 255 -- is cut in columns too: the one branch of
    the lookup no ordinary image reaches.  Against the independent 64-bit restatement, every pixel."""
    from effects_ref import defocus_by_summed_area_table
    rows, cols = 48, 8400
    rng = np.random.default_rng(4)
    orig = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    depth = rng.uniform(0, 255, (rows, cols)).astype(np.float32)
    depth[::7, ::331] = np.float32(40000.0)                          # k/2 far beyond the image: the whole (clipped) image, 8400 pixels wide
    depth[3::11, 5::977] = np.float32(900.0)                         # 3.5 x the nominal reach
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        art = up(np.zeros_like(orig))
        c.GPUSimulateDefocus(up(orig), up(depth), art, rows, cols)
        c.synchronize()
        assert c.get_option(rt.OPT_DEFOCUS_LAST_PATH) == 1
        got = down(art)
    want = defocus_by_summed_area_table(orig, depth)
    assert np.array_equal(got, want), f"{int((got != want).sum())} of {got.size} values differ"
