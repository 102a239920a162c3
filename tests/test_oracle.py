"""CPU tests that pin the oracle (no GPU): known-answer cases, the independent numpy
restatement (bit-exact, both FP-contraction variants), scipy's direct solve of the linear
system, and every small kernel's edge cases."""
import numpy as np
import pytest

import np_restatement as npr
from realtimedepthdiffusion_amd.synth import make_problem

f32 = np.float32


def test_lut_matches_formula_and_keeps_denormals(oracle, lut):
    assert lut.shape == (257,) and lut[0] == 1.0 and lut[256] == 0.0
    assert lut[255] > 0 and lut[255] < 1.2e-38          # denormal, not flushed
    assert np.all(np.diff(lut[:256]) < 0)
    ref = np.exp(-np.float64(f32(0.4)) * np.arange(256))
    assert np.allclose(lut[:219], ref[:219], rtol=3e-7)
    assert abs(lut[255] - 6e-45) < 2e-45


def test_omega_schedule_known_values(oracle):
    om = oracle.omega_schedule(400)
    assert np.all(om[:10] == 1.0)
    assert om[10] == f32(1.9609766) and om[11] == f32(1.9248844)
    assert abs(om[-1] - 1.7527453) < 1e-6
    assert np.array_equal(om, npr.omega_schedule(400))


def test_index_packing_known_answer(oracle):
    gray = np.array([[10, 13, 13], [20, 13, 250], [0, 255, 13]], np.uint8)
    idx = oracle.index_to_weight(gray, None, 0, 0)        # level == maxLevel: un-gated rule
    # centre pixel (1,1): left |13-20|=7, right |13-250|=237, up 0, down |13-255|=242
    assert idx[1, 1, 0] == 7 * 1000 + 237 and idx[1, 1, 1] == 0 * 1000 + 242
    # corner (0,0): no left/up -> 256
    assert idx[0, 0, 0] == 256 * 1000 + 3 and idx[0, 0, 1] == 256 * 1000 + 10
    assert idx[2, 2, 0] == 242 * 1000 + 256 and idx[2, 2, 1] == 237 * 1000 + 256


@pytest.mark.parametrize("level,max_level", [(0, 0), (0, 2), (1, 2), (2, 2)])
def test_index_maps_match_numpy(oracle, level, max_level):
    p = make_problem(37, 53, seed=5)
    rng = np.random.default_rng(7)
    depth = rng.uniform(0, 255, p["gray"].shape).astype(np.float32)
    depth[::5, ::3] = np.floor(depth[::5, ::3])
    idx = oracle.index_to_weight(p["gray"], depth, level, max_level)
    assert np.array_equal(idx, npr.pack_index(npr.index_maps(p["gray"], depth, level, max_level)))


def test_single_sweep_hand_computed(oracle, lut):
    # 1x3 image, flat gray (all weights 1), ends Dirichlet 0 and 90, centre free at 30, omega=1
    gray = np.full((1, 3), 7, np.uint8)
    mask = np.array([[255, 32, 255]], np.uint8)
    x = np.array([[0, 30, 90]], np.float32)
    idx = oracle.index_to_weight(gray, None, 0, 0)
    for contract in (0, 1):
        prev = np.zeros_like(x)
        out = oracle.sweep(x, idx, mask, prev, 1.0, lut, contract)
        # r = (0+90)/2 = 45 ; out = 1*(0.99*(45-30)+30-0)+0 = 44.85
        assert out[0, 0] == 0 and out[0, 2] == 90
        assert abs(out[0, 1] - 44.85) < 1e-5
        assert prev[0, 1] == 30 and prev[0, 0] == 0


def test_isolated_pixel_and_clamp(oracle, lut):
    # 1x1 image: no neighbours -> count == 0 -> r = 0 (src/GPUSolver.cu:103)
    gray = np.zeros((1, 1), np.uint8); mask = np.full((1, 1), 32, np.uint8)
    idx = oracle.index_to_weight(gray, None, 0, 0)
    x = np.array([[100.0]], np.float32); prev = np.zeros_like(x)
    out = oracle.sweep(x, idx, mask, prev, 1.0, lut, 0)
    assert abs(out[0, 0] - (0.99 * (0 - 100) + 100)) < 1e-4
    # clamp: neighbours at 300 -> mean clamps to 255 before relaxation
    gray = np.zeros((1, 2), np.uint8); mask = np.array([[32, 255]], np.uint8)
    idx = oracle.index_to_weight(gray, None, 0, 0)
    x = np.array([[10.0, 300.0]], np.float32); prev = np.zeros_like(x)
    out = oracle.sweep(x, idx, mask, prev, 1.0, lut, 0)
    assert abs(out[0, 0] - (0.99 * (255 - 10) + 10)) < 1e-4


def test_fma32_is_exact():
    rng = np.random.default_rng(0)
    a = rng.standard_normal(20000).astype(np.float32) * f32(1e3)
    b = rng.standard_normal(20000).astype(np.float32)
    c = (-(a.astype(np.float64) * b) * (1 + rng.standard_normal(20000) * 1e-7)).astype(np.float32)  # heavy cancellation
    got = npr.fma32(a, b, c)
    import fractions
    for i in range(0, 20000, 37):
        exact = fractions.Fraction(float(a[i])) * fractions.Fraction(float(b[i])) + fractions.Fraction(float(c[i]))
        # correctly rounded f32 of an exact rational: compare against both neighbours
        cand = np.float32(float(exact))
        lo, hi = np.nextafter(cand, f32(-np.inf)), np.nextafter(cand, f32(np.inf))
        best = min((cand, lo, hi), key=lambda v: abs(fractions.Fraction(float(v)) - exact))
        assert got[i] == best


@pytest.mark.parametrize("contract", [0, 1])
@pytest.mark.parametrize("shape,level,max_level", [((64, 64), 2, 2), ((33, 47), 1, 2), ((40, 40), 0, 2), ((1, 9), 0, 0), ((9, 1), 0, 0)])
def test_solver_matches_numpy_bitwise(oracle, lut, contract, shape, level, max_level):
    p = make_problem(shape[0], shape[1], seed=11 + shape[0])
    if (p["mask"] == 255).sum() == 0:
        p["mask"][0, 0] = 255; p["depth"][0, 0] = 64
    iters = 60
    want = npr.solve(p["depth"], p["mask"], p["gray"], iters, level, max_level, lut, contract)
    got = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], iters, level, max_level, lut, contract)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_contraction_variants_differ_but_are_close(oracle, lut):
    p = make_problem(64, 64, seed=3)
    a = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 200, 0, 0, lut, 0)
    b = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 200, 0, 0, lut, 1)
    d = np.abs(a - b).max()
    assert 0 < d < 1e-2        # recorded spread of "what nvcc might have produced"


def test_iteration_parity_and_zero_iterations(oracle, lut):
    p = make_problem(16, 16, seed=2)
    d0 = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 0, 0, 0, lut, 1)
    assert np.array_equal(d0, p["depth"])                  # maxIter = 0 returns the input
    for n in (1, 2, 3):
        got = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], n, 0, 0, lut, 1)
        want = npr.solve(p["depth"], p["mask"], p["gray"], n, 0, 0, lut, 1)
        assert np.array_equal(got, want)


def test_pitched_views_and_threads(oracle, lut):
    from realtimedepthdiffusion_amd.synth import pitched
    p = make_problem(48, 70, seed=9)
    dense = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 40, 0, 1, lut, 1)
    _, dv = pitched(p["depth"]); _, mv = pitched(p["mask"]); _, gv = pitched(p["gray"])
    got = oracle.solve(dv, mv, gv, 40, 0, 1, lut, 1, threads=4)
    assert np.array_equal(np.ascontiguousarray(got), dense)


def test_converges_to_direct_solution(oracle, lut):
    """Fixed point of the sweep == solution of (D - W) x = 0 with Dirichlet rows."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    p = make_problem(48, 48, seed=21)
    # low-contrast gray: with full-contrast edges (weights down to 1e-44) regions become numerically
    # disconnected and NO iterative scheme reaches the direct solution in a sane number of sweeps
    p["gray"] = np.ascontiguousarray(p["gray"] >> 4)
    rows, cols = p["gray"].shape
    maps = npr.index_maps(p["gray"], None, 0, 0)
    n = rows * cols
    pid = np.arange(n).reshape(rows, cols)
    free = (p["mask"] != 255)
    A = sp.lil_matrix((n, n)); b = np.zeros(n)
    off = {"left": (0, -1), "right": (0, 1), "up": (-1, 0), "down": (1, 0)}
    for y in range(rows):
        for x in range(cols):
            i = pid[y, x]
            if not free[y, x]:
                A[i, i] = 1.0; b[i] = p["depth"][y, x]; continue
            tot = 0.0
            for k, (dy, dx) in off.items():
                if maps[k][y, x] != 256:
                    w = float(lut[maps[k][y, x]]); A[i, pid[y + dy, x + dx]] = -w; tot += w
            A[i, i] = tot
    direct = spla.spsolve(A.tocsr(), b).reshape(rows, cols)
    got = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 4000, 0, 0, lut, 1)
    idx = oracle.index_to_weight(p["gray"], None, 0, 0)
    assert oracle.residual(got, idx, p["mask"], lut, 1) < 1e-3
    assert np.abs(got - direct).max() < 2e-2


def test_convert_pyrdown_paint_semantics(oracle):
    rng = np.random.default_rng(4)
    rows, cols = 9, 11
    edited = rng.integers(0, 255, (rows, cols, 3), dtype=np.uint8)
    mask = np.where(rng.random((rows, cols)) < 0.3, 255, 32).astype(np.uint8)
    dst = np.full((rows, cols), 255.0, np.float32)
    oracle.convert_to_float(edited, dst, mask)
    assert np.array_equal(dst, np.where(mask == 255, edited[..., 0].astype(np.float32), 255.0))
    # pyrDown: coarse (x,y) scans fine rows 2y-1,2y / cols 2x-1,2x; last hit (2x,2y) wins; never clears
    cr, cc = rows // 2, cols // 2
    cm = np.zeros((cr, cc), np.uint8); ce = np.zeros((cr, cc, 3), np.uint8)
    cm[0, 0] = 255                                  # stale coarse mark must survive
    oracle.pyrdown_annotation(mask, edited, cm, ce)
    for y in range(cr):
        for x in range(cc):
            hit = None
            for py in (2 * y - 1, 2 * y):
                for px in (2 * x - 1, 2 * x):
                    if 0 <= px < cols and 0 <= py < rows and mask[py, px] == 255:
                        hit = edited[py, px, 0]
            if hit is not None:
                assert cm[y, x] == 255 and ce[y, x, 0] == hit
            elif (y, x) != (0, 0):
                assert cm[y, x] == 0
            assert ce[y, x, 1] == 0 and ce[y, x, 2] == 0
    # paint: square brush, integer r/2
    e = np.zeros((rows, cols, 3), np.uint8); m = np.zeros((rows, cols), np.uint8)
    oracle.paint_image(5, 4, 192, 5, e, m)
    want = np.zeros((rows, cols), bool); want[2:7, 3:8] = True
    assert np.array_equal(m == 255, want) and np.all(e[want] == 192) and np.all(e[~want] == 0)
    oracle.paint_image(0, 0, 64, 3, e, m)            # clipped at the corner
    assert m[0, 0] == 255 and m[1, 1] == 255 and e[1, 1, 2] == 64


def test_effects_known_answers(oracle):
    rows, cols = 6, 8
    rng = np.random.default_rng(8)
    orig = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    gray = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
    depth = np.zeros((rows, cols), np.float32)
    for c in (0, 1):
        assert np.array_equal(oracle.desaturate(orig, gray, depth, c), orig)         # f = 0 -> original
        assert np.array_equal(oracle.haze(orig, depth, c), orig)                     # t = 1 -> original
    depth[:] = 255
    assert np.array_equal(oracle.desaturate(orig, gray, depth, 1), np.repeat(gray[..., None], 3, 2))   # f = 1 -> gray
    t = np.exp(np.float32(-2.0))
    want = (t * orig.astype(np.float32) + (1 - t) * 255).astype(np.float64)
    got = oracle.haze(orig, depth, 0).astype(np.float64)
    assert np.abs(got - np.floor(want)).max() <= 1
    # defocus: K = int(0.025*sqrt(36+64)) = 0 -> k = 0 -> count == 0 -> copy
    assert np.array_equal(oracle.defocus(orig, depth), orig)
    # larger image so K >= 2: 80x80 -> K = int(0.025*113.1) = 2 ; depth 255 -> k = 2 -> window [y-1,y+1) x [x-1,x+1)
    orig = rng.integers(0, 256, (80, 80, 3), dtype=np.uint8)
    depth = np.full((80, 80), 255, np.float32)
    got = oracle.defocus(orig, depth)
    y, x = 10, 20
    want = orig[y - 1:y + 1, x - 1:x + 1].reshape(-1, 3).astype(np.float32).sum(0) / 4
    assert np.array_equal(got[y, x], want.astype(np.uint8))
    want0 = orig[0:1, 0:1].reshape(-1, 3).astype(np.float32).sum(0) / 1      # clipped window at the corner
    assert np.array_equal(got[0, 0], want0.astype(np.uint8))


def test_cascade_helpers(oracle):
    bgr = np.zeros((2, 2, 3), np.uint8); bgr[0, 0] = (255, 255, 255); bgr[0, 1] = (255, 0, 0); bgr[1, 0] = (0, 255, 0); bgr[1, 1] = (0, 0, 255)
    g = oracle.bgr2gray(bgr)
    assert g.tolist() == [[255, 29], [150, 76]]
    flat = np.full((7, 9), 77, np.uint8)
    d = oracle.pyrdown_u8(flat)
    assert d.shape == (4, 5) and np.all(d == 77)
    up = oracle.pyrup_f32(np.full((4, 5), 3.5, np.float32), 7, 9)
    assert up.shape == (7, 9) and np.all(up == 3.5)
    # cv::pyrUp with an explicit odd size (imgproc pyramids.cpp, pyrUp_), worked by hand on [[1,2,4],[8,16,32]] -> 5x7: row buffers
    # 10,12,17,24,30,32,32 and 80,96,136,192,240,256,256 (x=0: 6 s0 + 2 s1; x=n-1: s[n-2] + 7 s[n-1], then 8 s[n-1]; the 7th
    # column repeats the 6th); rows: mirror above (r1 r0 r1), replicate below (r0 r1 r1), the 5th row repeats the 3rd
    s = np.array([[1, 2, 4], [8, 16, 32]], np.float32)
    b0 = np.array([10, 12, 17, 24, 30, 32, 32], np.float32); b1 = np.array([80, 96, 136, 192, 240, 256, 256], np.float32)
    want = np.stack([(b1 + 6 * b0 + b1) / 64, (b0 + b1) * 4 / 64, (b0 + 6 * b1 + b1) / 64, (b1 + b1) * 4 / 64, (b0 + 6 * b1 + b1) / 64])
    assert np.array_equal(oracle.pyrup_f32(s, 5, 7), want)
    # cv::cuda::pyrUp for exact doubling: the same numbers (mirror top/left, replicate bottom/right), here without contraction
    assert np.array_equal(oracle.pyrup_f32(s, 4, 6, contract=0), want[:4, :6])
    assert oracle.pyrup_f32(s, 4, 6, contract=0)[0, 0] == 3.4375 and oracle.pyrup_f32(s, 4, 6, contract=0)[3, 5] == 32.0   # not reflect-101: that gives 24.25 at [3,5]
    one = oracle.pyrup_f32(np.array([[3.0]], np.float32), 3, 3)
    assert np.all(one == 3.0)
    ramp = np.tile(np.arange(6, dtype=np.float32), (4, 1))
    up = oracle.pyrup_f32(ramp, 8, 12)
    assert np.allclose(up[3, 2:10], np.arange(2, 10) / 2.0)          # linear ramps are reproduced in the interior
    assert oracle.depth_to_u8(np.array([[0.5, 1.5, 2.5, -3, 300, 254.5]], np.float32)).tolist() == [[0, 2, 2, 0, 255, 254]]


def test_summed_area_restatement_of_defocus_agrees_with_the_literal_gather(oracle):
    """tests/effects_ref.py (used for the 4K / 8K defocus tests on the GPU) against orc_defocus, every pixel, and orc_defocus_at
    (the literal gather for listed pixels) against orc_defocus."""
    from effects_ref import defocus_by_summed_area_table, effect_inputs
    orig, depth = effect_inputs(300, 420, 3)
    lit = oracle.defocus(orig, depth, threads=4)
    assert np.array_equal(defocus_by_summed_area_table(orig, depth), lit)
    ys = np.array([0, 10, 299, 150, 77]); xs = np.array([0, 400, 419, 200, 5])
    assert np.array_equal(oracle.defocus_at(orig, depth, ys, xs), lit[ys, xs])


def test_deterministic_exp_of_the_haze_restatement(oracle):
    """orc_expf_det (the exp the haze restatement and the GPU kernel share: a fixed f64 operation sequence, no libm): equal to the
    correctly rounded f32 of numpy's f64 exp on 3 M samples, special values as IEEE wants them, and different from this host's libm expf on
    well under 0.2 % of the samples (glibc's expf is NOT correctly rounded: it misrounds ~0.06 % of arguments)."""
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-3, 0.6, 2_000_000), rng.uniform(-110, 95, 1_000_000),
                        np.array([0, -0.0, 1e-40, -1e-40, 88.7, -103.9, -87.5, -100.0])]).astype(np.float32)
    v, n_differ = oracle.expf_det(x)
    with np.errstate(all="ignore"):
        ref = np.exp(x.astype(np.float64)).astype(np.float32)
    assert np.array_equal(v, ref)
    assert n_differ < 0.002 * x.size                             # glibc's expf misrounds ~0.06 % of arguments; this one none of these
    sp, _ = oracle.expf_det(np.array([np.nan, np.inf, -np.inf, 89.5, -104.5], np.float32))
    assert np.isnan(sp[0]) and sp[1] == np.inf and sp[2] == 0 and sp[3] == np.inf and sp[4] == 0


def test_cascade_pieces_agree_with_scipy_witnesses(oracle):
    """The OpenCV steps of the estimate are third-party arithmetic restated from published formulas (oracle/rtdd_cascade_oracle.c) and
    unpinned by the reference.  Independent witnesses, written against the formulas and not against that file: BT.601 fixed-point gray in
    numpy integers; pyrDown as scipy's separable correlation with [1 4 6 4 1] in `mirror` mode (= reflect-101) on integers, rounded
    (s + 128) >> 8, every second sample; the exact-doubling pyrUp as zero insertion + the same kernel x 4 / 256 in f64 (interior only:
    the border rule is OpenCV's own), within f32 rounding."""
    from scipy import ndimage
    rng = np.random.default_rng(11)
    for rows, cols in ((7, 9), (67, 120), (135, 241), (270, 480)):
        bgr = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
        g = oracle.bgr2gray(bgr)
        b, gr, r = (bgr[..., i].astype(np.int64) for i in range(3))
        assert np.array_equal(g, ((b * 1868 + gr * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8))
        k = np.array([1, 4, 6, 4, 1], np.int64)
        full = ndimage.correlate1d(ndimage.correlate1d(g.astype(np.int64), k, axis=0, mode="mirror"), k, axis=1, mode="mirror")
        assert np.array_equal(oracle.pyrdown_u8(g), ((full[::2, ::2] + 128) >> 8).astype(np.uint8)), (rows, cols)
        src = rng.uniform(0, 255, (rows, cols)).astype(np.float32)
        up = oracle.pyrup_f32(src, 2 * rows, 2 * cols, contract=0)
        z = np.zeros((2 * rows, 2 * cols)); z[::2, ::2] = src
        kf = np.array([1, 4, 6, 4, 1], np.float64)
        want = ndimage.correlate1d(ndimage.correlate1d(z, kf, axis=0, mode="constant"), kf, axis=1, mode="constant") * 4 / 256
        assert np.abs(up[2:-3, 2:-3] - want[2:-3, 2:-3]).max() <= 2e-4, (rows, cols)
