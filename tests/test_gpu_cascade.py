"""Whole-estimate driver on the GPU against the oracle-composed cascade (-m gpu)."""
import ctypes as C

import numpy as np
import pytest

import realtimedepthdiffusion_amd as rt
from cascade_ref import Cascade, pyramid_levels
from golden_util import NAMES, load
from gpu_util import assert_bit_equal, down, up
from realtimedepthdiffusion_amd.synth import make_problem

pytestmark = pytest.mark.gpu


def _bgr(rows, cols, seed):
    p = make_problem(rows, cols, seed=seed)
    g = p["gray"].astype(np.int32)
    rng = np.random.default_rng(seed)
    bgr = np.stack([np.clip(g + rng.integers(-20, 21, g.shape), 0, 255) for _ in range(3)], -1).astype(np.uint8)
    ann = np.where(p["mask"] == 255, p["edited"][..., 0], 32).astype(np.uint8)
    ann[ann == 32 + 0] = 32
    return bgr, ann


@pytest.mark.parametrize("shape", [(6, 8), (67, 120), (135, 241), (853, 1280), (1, 7), (7, 1), (2, 2)])
def test_third_party_pieces_match_the_restatement(oracle, shape):
    rows, cols = shape
    bgr, _ = _bgr(rows, cols, 3)
    with rt.Context(0) as c:
        g = up(np.zeros((rows, cols), np.uint8))
        c.bgr2gray(up(bgr), g, rows, cols)
        gray = oracle.bgr2gray(bgr)
        assert np.array_equal(down(g), gray)
        d = up(np.zeros(((rows + 1) // 2, (cols + 1) // 2), np.uint8))
        c.pyrdown_gray(g, rows, cols, d)
        assert np.array_equal(down(d), oracle.pyrdown_u8(gray))
        src = np.random.default_rng(1).uniform(-10, 300, (rows, cols)).astype(np.float32)
        for contract in (1, 0):                                         # the exact-doubling branch is cv::cuda::pyrUp: contraction matters there
            c.set_option(rt.OPT_FP_CONTRACT, contract)
            for drows, dcols in ((2 * rows, 2 * cols), (2 * rows + 1, 2 * cols + 1), (2 * rows + 1, 2 * cols), (2 * rows, 2 * cols + 1)):
                dst = up(np.zeros((drows, dcols), np.float32))
                c.pyrup_depth(up(src), rows, cols, dst, drows, dcols)
                assert_bit_equal(down(dst), oracle.pyrup_f32(src, drows, dcols, contract=contract), f"pyrUp {shape}->{drows}x{dcols} contract {contract}")
        c.set_option(rt.OPT_FP_CONTRACT, 1)
        u = up(np.zeros((rows, cols), np.uint8))
        vals = src.copy(); vals[0, :6] = [0.5, 1.5, 2.5, 254.5, 255.5, -0.5][:min(6, cols)] if cols >= 6 else vals[0, :6]
        c.depth_to_u8(up(vals), u, rows, cols)
        assert np.array_equal(down(u), oracle.depth_to_u8(vals))


def test_pyramid_level_rule():
    for rows, cols in ((256, 256), (1080, 1920), (2160, 3840), (4320, 7680), (624, 672), (44, 100), (90, 91)):
        assert rt.lib().rtdd_pyramid_levels(rows, cols) == pyramid_levels(rows, cols)
    assert pyramid_levels(1080, 1920) == 5 and pyramid_levels(2160, 3840) == 6 and pyramid_levels(4320, 7680) == 7


@pytest.mark.parametrize("shape,iters", [((256, 256), 1000), ((135, 241), 300), ((297, 211), 200), ((90, 91), 50)])
def test_estimate_matches_oracle_cascade(oracle, lut, shape, iters):
    rows, cols = shape
    bgr, ann = _bgr(rows, cols, 11)
    ref = Cascade(oracle, bgr, ann, lut, 1, threads=min(8, oracle.max_threads()))
    ref.estimate(iters)
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        P = c.pyramid_create(rows, cols)
        assert P == ref.P
        c.pyramid_set_image(up(bgr)); c.pyramid_set_annotation(up(ann))
        c.estimate_depth(iters)
        c.synchronize()
        for l in range(P):
            assert np.array_equal(c.pyramid_download(rt.IMG_GRAY, l), ref.gray[l]), f"gray {l}"
            assert np.array_equal(c.pyramid_download(rt.IMG_SCRIBBLE, l), ref.scribble[l]), f"scribble {l}"
            assert np.array_equal(c.pyramid_download(rt.IMG_EDITED, l)[..., 0], ref.edited[l][..., 0]), f"edited {l}"
        for l in range(P - 1, -1, -1):
            assert_bit_equal(c.pyramid_download(rt.IMG_DEPTH, l), ref.depth[l], f"depth level {l}")
        assert np.array_equal(c.pyramid_download(rt.IMG_DEPTH_U8), ref.depth_u8)
        # warm start (--live): a second estimate continues from the first one's coarsest solution
        ref.estimate(iters)
        c.estimate_depth(iters); c.synchronize()
        assert_bit_equal(c.pyramid_download(rt.IMG_DEPTH, 0), ref.depth[0], "second estimate (warm start)")


@pytest.mark.parametrize("name", NAMES)
def test_estimate_reproduces_golden_crops(name):
    """End to end from the committed decoded crop: gray, annotation decode, pyramid, solves, pyrUp, u8."""
    g = load(name)
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        assert c.pyramid_create(256, 256) == 3
        c.pyramid_set_image(up(g["bgr"])); c.pyramid_set_annotation(up(g["annotation"]))
        c.estimate_depth(1000); c.synchronize()
        for l in range(3):
            assert np.array_equal(c.pyramid_download(rt.IMG_GRAY, l), g[f"gray{l}"])
            assert_bit_equal(c.pyramid_download(rt.IMG_DEPTH, l), g[f"depth_after_c1_L{l}"], f"{name} level {l}")
        assert np.array_equal(c.pyramid_download(rt.IMG_DEPTH_U8), g["depth_u8"])


def test_estimate_1080p_full_cascade(oracle, lut):
    """Full-size: the 5-level 1080p cascade (250 Mpixel-iterations) against the oracle on the host cores."""
    rows, cols = 1080, 1920
    bgr, ann = _bgr(rows, cols, 1234)
    ref = Cascade(oracle, bgr, ann, lut, 1, threads=oracle.max_threads())
    ref.estimate(1000)
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        assert c.pyramid_create(rows, cols) == 5
        c.pyramid_set_image(up(bgr)); c.pyramid_set_annotation(up(ann))
        c.estimate_depth(1000); c.synchronize()
        got = c.pyramid_download(rt.IMG_DEPTH, 0)
        assert np.abs(got - ref.depth[0]).max() <= 1e-4
        assert_bit_equal(got, ref.depth[0], "1080p cascade")
        assert np.array_equal(c.pyramid_download(rt.IMG_DEPTH_U8), ref.depth_u8)


@pytest.mark.parametrize("rows,cols,levels,seed", [(2160, 3840, 6, 1234), (4320, 7680, 7, 1234)])
def test_estimate_4k_and_8k_full_cascades(oracle, lut, rows, cols, levels, seed):
    """src/main.cpp:95,261-288 at 3840x2160 (P = 6: 507 Mpixel-iterations) and 7680x4320 (P = 7: 1005): every level of the
    cascade bit for bit against the oracle cascade on the host cores.  The coarsest level of both is odd-sized (67 rows under a
    135-row level): the host-pyrUp branch and the ceil/floor gray quirk sit inside."""
    bgr, ann = _bgr(rows, cols, seed)
    ref = Cascade(oracle, bgr, ann, lut, 1, threads=oracle.max_threads())
    ref.estimate(1000)
    assert ref.P == levels
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        assert c.pyramid_create(rows, cols) == levels
        c.pyramid_set_image(up(bgr)); c.pyramid_set_annotation(up(ann))
        c.estimate_depth(1000); c.synchronize()
        for l in range(levels):
            assert np.array_equal(c.pyramid_download(rt.IMG_GRAY, l), ref.gray[l]), f"gray {l}"
            assert np.array_equal(c.pyramid_download(rt.IMG_SCRIBBLE, l), ref.scribble[l]), f"scribble {l}"
        for l in range(levels - 1, -1, -1):
            got = c.pyramid_download(rt.IMG_DEPTH, l)
            assert np.abs(got - ref.depth[l]).max() <= 1e-4
            assert_bit_equal(got, ref.depth[l], f"{cols}x{rows} cascade, level {l}")
        assert np.array_equal(c.pyramid_download(rt.IMG_DEPTH_U8), ref.depth_u8)
        ref.estimate(1000)                                              # --live: the warm-started second frame
        c.estimate_depth(1000); c.synchronize()
        assert_bit_equal(c.pyramid_download(rt.IMG_DEPTH, 0), ref.depth[0], f"{cols}x{rows} cascade, second (warm-started) estimate")


def test_estimate_heals_a_timed_out_persistent_level(oracle, lut, capfd):
    """rtdd_estimate_depth with one tile's hand-off flag withheld (RTDD_OPT_DEBUG_WITHHOLD_TILE): the first persistent level of the
    cascade times out, every copy-back behind it stores nothing, and the synchronising call runs the cascade again from that level
    on without persistence -- every level's depth image and the u8 map are the oracle's, as if nothing had happened; the second,
    warm-started estimate too (src/GPUSolver.cu:311-314: the reference's solver always leaves a valid depth map)."""
    rows, cols = 1080, 1920
    bgr, ann = _bgr(rows, cols, 77)
    ref = Cascade(oracle, bgr, ann, lut, 1, threads=oracle.max_threads())
    ref.estimate(1000)
    capfd.readouterr()
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        levels = c.pyramid_create(rows, cols)
        c.pyramid_set_image(up(bgr)); c.pyramid_set_annotation(up(ann))
        c.set_option(rt.OPT_DEBUG_POLL_LIMIT_US, 3000); c.set_option(rt.OPT_DEBUG_WITHHOLD_TILE, 1)
        c.estimate_depth(1000); c.synchronize()
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 1 and c.get_option(rt.OPT_PERSISTENT) == 0
        assert capfd.readouterr().err.count("rtdd: persistent sweep kernel") == 1
        for l in range(levels - 1, -1, -1):
            assert_bit_equal(c.pyramid_download(rt.IMG_DEPTH, l), ref.depth[l], f"healed cascade, level {l}")
        assert np.array_equal(c.pyramid_download(rt.IMG_DEPTH_U8), ref.depth_u8)
        ref.estimate(1000)
        c.estimate_depth(1000); c.synchronize()
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 1
        assert_bit_equal(c.pyramid_download(rt.IMG_DEPTH, 0), ref.depth[0], "estimate after a healed one (warm start)")


class _Pageable:
    """an ordinary numpy array where the tests otherwise pass a page-locked host image"""
    def __init__(self, shape):
        self.a = np.zeros(shape, np.uint8)


@pytest.mark.parametrize("out_kind", ["page_locked", "page_locked_direct", "page_locked_staged", "pageable"])
@pytest.mark.parametrize("withhold", [False, True])
def test_live_frames_pipelined_match_the_oracle(oracle, lut, withhold, out_kind):
    """rtdd_live_submit / rtdd_live_wait (src/main.cpp:232-295 per frame: upload scribble + edited, estimate, download the u8 map), two
    frames in flight on two streams, page-locked host images: every frame's map is the oracle's n-th warm-started estimate -- also when
    the first frame's persistent launch times out and both frames in flight are healed behind the caller's back.  Two ways for the map
    to reach the host -- stored by the copy-back kernel straight into a page-locked buffer, or staged on the device and downloaded by
    rtdd_live_wait -- chosen per frame by default (direct when no other frame is in flight), forced either way by the option, and staged
    whatever the option says when the buffer is an ordinary allocation."""
    rows, cols = 540, 960
    bgr, ann = _bgr(rows, cols, 31)
    ref = Cascade(oracle, bgr, ann, lut, 1, threads=oracle.max_threads())
    with rt.Context(0) as c:
        import torch
        c.set_stream(torch.cuda.Stream().cuda_stream if not withhold else 0)        # a stream of its own, and the null stream
        c.GPULoadWeights(0.4)
        c.pyramid_create(rows, cols)
        c.pyramid_set_image(up(bgr)); c.pyramid_set_annotation(up(ann)); c.synchronize()
        scr = rt.host_image((rows, cols)); ed = rt.host_image((rows, cols, 3))
        out = [(_Pageable if out_kind == "pageable" else rt.host_image)((rows, cols)) for _ in range(2)]
        assert c.get_option(rt.OPT_LIVE_ZERO_COPY) == 1
        if out_kind in ("page_locked_staged", "page_locked_direct"):
            c.set_option(rt.OPT_LIVE_ZERO_COPY, 0 if out_kind == "page_locked_staged" else 2)
        scr.a[...] = ref.scribble[0]; ed.a[...] = ref.edited[0]
        if withhold:
            c.set_option(rt.OPT_DEBUG_POLL_LIMIT_US, 3000); c.set_option(rt.OPT_DEBUG_WITHHOLD_TILE, 1)
        frames = 5
        got = []
        for n in range(frames):
            if n >= 2:
                c.live_wait(); got.append(out[n % 2].a.copy())
            c.live_submit(scr.a, ed.a, out[n % 2].a, 1000)
            assert c.live_pending() == min(n + 1, 2)
        while c.live_pending():
            k = len(got)
            c.live_wait(); got.append(out[k % 2].a.copy())
        c.synchronize()
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) == (1 if withhold else 0)
        for n in range(frames):
            ref.estimate(1000)
            assert np.array_equal(got[n], ref.depth_u8), f"frame {n}"
        assert_bit_equal(c.pyramid_download(rt.IMG_DEPTH, 0), ref.depth[0], "depth after the last live frame")
        # a frame without an upload: the annotation on the device stays
        c.live_submit(None, None, out[0].a, 1000); c.live_wait()
        ref.estimate(1000)
        assert np.array_equal(out[0].a, ref.depth_u8)
        with pytest.raises(rt.RtddError):
            c.live_wait()                                                           # nothing in flight


@pytest.mark.parametrize("width,host_pitch", [(910, 910), (2730, 2730), (517, 517), (33, 33), (910, 912), (911, 915), (1, 1), (64, 64)])
def test_upload_and_download_with_any_host_pitch(width, host_pitch):
    """rtdd_upload / rtdd_download move a host image whose pitch is no multiple of four through a contiguous device buffer (the runtime's 2-D
    copy takes ~9 us per ROW then: 8 ms for the dataset's 910-pixel-wide Arara); whatever the route, the bytes arrive, and only they."""
    import torch
    rows = 37
    rng = np.random.default_rng(width)
    host = rng.integers(0, 256, (rows, host_pitch), dtype=np.uint8)
    with rt.Context(0) as c:
        dev = torch.full((rows + 2, 1024 if width <= 1024 else 3072), 7, dtype=torch.uint8, device="cuda:0")
        dp = dev.stride(0)
        c._check(rt.lib().rtdd_upload(c._h, C.c_void_p(dev[1].data_ptr()), C.c_size_t(dp), C.c_void_p(host.ctypes.data), C.c_size_t(host_pitch), C.c_size_t(width), C.c_int(rows)))
        got = dev.cpu().numpy()
        assert np.array_equal(got[1:1 + rows, :width], host[:, :width])
        assert (got[0] == 7).all() and (got[-1] == 7).all() and (got[1:1 + rows, width:] == 7).all()
        back = np.full((rows, host_pitch), 9, np.uint8)
        c._check(rt.lib().rtdd_download(c._h, C.c_void_p(back.ctypes.data), C.c_size_t(host_pitch), C.c_void_p(dev[1].data_ptr()), C.c_size_t(dp), C.c_size_t(width), C.c_int(rows)))
        assert np.array_equal(back[:, :width], host[:, :width]) and (back[:, width:] == 9).all()


@pytest.mark.parametrize("pageable", [False, True])
def test_live_frames_of_an_odd_width(oracle, lut, pageable):
    """Pipelined live frames of an image whose width is no multiple of four (contiguous host images: pitches 241 and 723): uploads and the
    staged download take the contiguous route; every map is the oracle's."""
    rows, cols = 135, 241
    bgr, ann = _bgr(rows, cols, 12)
    ref = Cascade(oracle, bgr, ann, lut, 1, threads=4)
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        c.pyramid_create(rows, cols)
        c.pyramid_set_image(up(bgr)); c.pyramid_set_annotation(up(ann)); c.synchronize()
        mk = _Pageable if pageable else rt.host_image
        scr = mk((rows, cols)); ed = mk((rows, cols, 3)); out = [mk((rows, cols)) for _ in range(2)]
        scr.a[...] = ref.scribble[0]; ed.a[...] = ref.edited[0]
        got = []
        for n in range(4):
            if n >= 2:
                c.live_wait(); got.append(out[n % 2].a.copy())
            c.live_submit(scr.a, ed.a, out[n % 2].a, 1000)
        while c.live_pending():
            k = len(got); c.live_wait(); got.append(out[k % 2].a.copy())
        for n in range(4):
            ref.estimate(1000)
            assert np.array_equal(got[n], ref.depth_u8), f"frame {n}"


@pytest.mark.parametrize("zero_copy", [1, 2])
def test_live_frame_into_a_window_of_a_wider_page_locked_image(oracle, lut, zero_copy):
    """The host's map may be a window of a larger page-locked image (a pitch of its own, an interior first pixel): the copy-back kernel
    stores into exactly that window -- the pixels around it keep their values."""
    rows, cols = 135, 241
    bgr, ann = _bgr(rows, cols, 8)
    ref = Cascade(oracle, bgr, ann, lut, 1, threads=4)
    ref.estimate(1000)
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        c.pyramid_create(rows, cols)
        c.pyramid_set_image(up(bgr)); c.pyramid_set_annotation(up(ann)); c.synchronize()
        c.set_option(rt.OPT_LIVE_ZERO_COPY, zero_copy)
        with pytest.raises(rt.RtddError):
            c.set_option(rt.OPT_LIVE_ZERO_COPY, 3)
        scr = rt.host_image((rows, cols)); ed = rt.host_image((rows, cols, 3)); big = rt.host_image((rows + 6, cols + 37))
        scr.a[...] = ref.scribble[0]; ed.a[...] = ref.edited[0]; big.a[...] = 77
        win = big.a[3:3 + rows, 19:19 + cols]
        c.live_submit(scr.a, ed.a, win, 1000); c.live_wait()
        assert np.array_equal(win, ref.depth_u8)
        frame = big.a.copy(); frame[3:3 + rows, 19:19 + cols] = 77
        assert (frame == 77).all()


@pytest.mark.parametrize("rows,cols,lds", [(270, 480, 1), (270, 480, 0), (333, 517, 1), (333, 517, 0), (67, 120, 1), (40, 33, 1)])
def test_annotation_pyramid_follows_every_write(oracle, lut, rows, cols, lds):
    """(Both forms of the one-launch pyramid kernel: levels walked in LDS / read back from global memory -- RTDD_OPT_ANNOTATION_LDS.)
    The coarse annotation levels are rebuilt by the first estimate after the annotation changed, not by every estimate
    (src/main.cpp:249-259 does it every time; K6 only ever adds, so the images are the same).  Every library call that writes the
    annotation must be noticed: rtdd_paint_image on the pyramid's own level-0 images, rtdd_upload, set_annotation; and a caller
    that writes through the raw pointers says so (rtdd_pyramid_annotation_changed)."""
    bgr, ann = _bgr(rows, cols, 5)
    ref = Cascade(oracle, bgr, ann, lut, 1, threads=4)
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        c.set_option(rt.OPT_ANNOTATION_LDS, lds)
        levels = c.pyramid_create(rows, cols)
        c.pyramid_set_image(up(bgr)); c.pyramid_set_annotation(up(ann))

        def check(what):
            ref.estimate(1000); c.estimate_depth(1000); c.synchronize()
            for l in range(levels):
                assert np.array_equal(c.pyramid_download(rt.IMG_SCRIBBLE, l), ref.scribble[l]), f"{what}: scribble {l}"
                assert np.array_equal(c.pyramid_download(rt.IMG_EDITED, l)[..., 0], ref.edited[l][..., 0]), f"{what}: edited {l}"
                assert_bit_equal(c.pyramid_download(rt.IMG_DEPTH, l), ref.depth[l], f"{what}: depth {l}")

        check("cold"); check("unchanged annotation")
        sp, spitch, _, _ = c.pyramid_image(rt.IMG_SCRIBBLE, 0); ep, epitch, _, _ = c.pyramid_image(rt.IMG_EDITED, 0)
        oracle.paint_image(cols // 5, rows // 3, 64, 11, ref.edited[0], ref.scribble[0])
        c.GPUPaintImage(cols // 5, rows // 3, 64, 11, (ep, epitch), (sp, spitch), rows, cols)
        check("after rtdd_paint_image")
        def upload(ctx_, host, ptr, pitch, width):
            host = np.ascontiguousarray(host)
            ctx_._check(rt.lib().rtdd_upload(ctx_._h, C.c_void_p(ptr), C.c_size_t(pitch), C.c_void_p(host.ctypes.data), C.c_size_t(width), C.c_size_t(width), C.c_int(rows)))

        oracle.paint_image(cols * 5 // 8, rows * 3 // 4, 192, 15, ref.edited[0], ref.scribble[0])
        upload(c, ref.scribble[0], sp, spitch, cols); upload(c, ref.edited[0], ep, epitch, cols * 3)
        check("after rtdd_upload into the pyramid's images")
        # through the raw pointers, behind this context's back (another context's copy): the caller has to say so
        oracle.paint_image(cols // 12, rows // 9, 0, 21, ref.edited[0], ref.scribble[0])
        with rt.Context(0) as other:
            upload(other, ref.scribble[0], sp, spitch, cols); upload(other, ref.edited[0], ep, epitch, cols * 3)
        c.pyramid_annotation_changed()
        check("after a foreign write + rtdd_pyramid_annotation_changed")


@pytest.mark.parametrize("name", NAMES[:2])
def test_refine_depth_converges_the_estimate(oracle, name):
    """rtdd_refine_depth (extension): estimate, then SOR cycles on the finest level to a 1e-4 residual -- the same sweep
    count, residual and bits as the schedule restated over the oracle, starting from the estimate's own depth (whose
    values gate the level-0 weights, src/GPUSolver.cu:188-218)."""
    from test_gpu_parity import _sor_cycles_restated
    g = load(name)
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        c.pyramid_create(256, 256)
        c.pyramid_set_image(up(g["bgr"])); c.pyramid_set_annotation(up(g["annotation"]))
        c.estimate_depth(1000); c.synchronize()
        start = c.pyramid_download(rt.IMG_DEPTH, 0)
        its, res = c.refine_depth(method=rt.METHOD_RED_BLACK_GS, tolerance=1e-4, relaxation=rt.RELAXATION_AUTO)
        got = c.pyramid_download(rt.IMG_DEPTH, 0)
        idx = oracle.index_to_weight(g["gray0"], start, 0, 2)
        x = start.copy()
        want_its, want_res = _sor_cycles_restated(oracle, x, idx, g["mask0"], g["lut"], 1, 1e-4, 200000)
        assert (its, res) == (want_its, np.float32(want_res)) and res <= 1e-4
        assert_bit_equal(got, x, "refined depth")
        assert np.array_equal(c.pyramid_download(rt.IMG_DEPTH_U8), oracle.depth_to_u8(x))


def test_full_size_dataset_pair_end_to_end(oracle, lut):
    """A bundled image/annotation pair at its own resolution (tests/golden/Dog_full.npz: 672x624, 4 pyramid levels, stored
    decoded): the whole estimate -- gray pyramid, annotation decode and pyramid, four solves, pyrUp + re-injection, u8 --
    and the three effects, GPU against the oracle computed here, plus the hashes recorded when the fixture was made."""
    import os
    from golden_util import GOLDEN_DIR, sha
    g = np.load(os.path.join(GOLDEN_DIR, "Dog_full.npz"), allow_pickle=False)
    bgr, ann = g["bgr"], g["annotation"]
    rows, cols = bgr.shape[:2]
    ref = Cascade(oracle, bgr, ann, lut, 1, threads=min(8, oracle.max_threads()))
    ref.estimate(1000)
    assert ref.P == int(g["levels"]) and sha(ref.depth_u8) == str(g["depth_u8_sha"]) and sha(ref.depth[0]) == str(g["depth_sha"])
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        assert c.pyramid_create(rows, cols) == ref.P
        c.pyramid_set_image(up(bgr)); c.pyramid_set_annotation(up(ann))
        c.estimate_depth(1000); c.synchronize()
        for l in range(ref.P - 1, -1, -1):
            assert_bit_equal(c.pyramid_download(rt.IMG_DEPTH, l), ref.depth[l], f"Dog full size, level {l}")
        assert np.array_equal(c.pyramid_download(rt.IMG_DEPTH_U8), ref.depth_u8)
        o = up(bgr); d = up(ref.depth[0]); gray = up(ref.gray[0]); art = up(np.zeros_like(bgr))
        c.GPUSimulateDesaturation(o, gray, d, art, rows, cols)
        assert np.array_equal(down(art), oracle.desaturate(bgr, ref.gray[0], ref.depth[0], 1))
        c.GPUSimulateDefocus(o, d, art, rows, cols)
        assert np.array_equal(down(art), oracle.defocus(bgr, ref.depth[0], threads=min(8, oracle.max_threads())))
        c.GPUSimulateHaze(o, d, art, rows, cols)
        assert np.array_equal(down(art), oracle.haze(bgr, ref.depth[0], 1))          # bit-exact since round 3: one deterministic exp on both sides


def test_the_spelt_out_cascade_heals_too(oracle, lut):
    """the cascade spelt out through the C ABI -- solve(level 1) -> rtdd_pyrup_depth -> rtdd_convert_to_float -> solve(level 0)
    -> rtdd_depth_to_u8, all queued without a synchronisation -- with level 1's persistent launch timing out.  k_finish stores nothing,
    so rtdd_pyrup_depth would read level 1's INPUT and the heal would replay only the solves: a silently wrong map.  The calls that read
    a solve's output without being logged now settle the log first; the result is the oracle's."""
    rows, cols = 2160, 3840                                    # level 1 = 1080p: persistent (252 tiles); level 0 = 4K: launch per block
    p0 = make_problem(rows, cols, seed=61)
    r1, c1 = rows // 2, cols // 2
    p1 = make_problem(r1, c1, seed=62)
    # reference: level 1 (gated rule, level 1 of 2 is the coarsest -> un-gated) then pyrUp, injection, level 0, u8
    d1 = oracle.solve(p1["depth"].copy(), p1["mask"], p1["gray"], 40, 1, 1, lut, 1, threads=oracle.max_threads())
    d0 = oracle.pyrup_f32(d1, rows, cols, contract=1)
    oracle.convert_to_float(p0["edited"], d0, p0["mask"])
    d0 = oracle.solve(d0, p0["mask"], p0["gray"], 16, 0, 1, lut, 1, threads=oracle.max_threads())
    want_u8 = oracle.depth_to_u8(d0)
    with rt.Context(0) as c:
        c.GPUAllocateDeviceMemory(rows, cols, 2); c.GPULoadWeights(0.4)
        c.set_option(rt.OPT_DEBUG_POLL_LIMIT_US, 3000); c.set_option(rt.OPT_DEBUG_WITHHOLD_TILE, 100 + 1)
        g1, m1, x1 = up(p1["gray"]), up(p1["mask"]), up(p1["depth"])
        g0, m0, e0 = up(p0["gray"]), up(p0["mask"]), up(p0["edited"])
        x0 = up(np.zeros((rows, cols), np.float32)); u8 = up(np.zeros((rows, cols), np.uint8))
        c.GPUMatrixFreeSolver(x1, m1, g1, r1, c1, 0.4, 40, 0.0, 1)
        c.pyrup_depth(x1, r1, c1, x0, rows, cols)
        c.GPUConvertToFloat(e0, x0, m0, rows, cols)
        c.GPUMatrixFreeSolver(x0, m0, g0, rows, cols, 0.4, 16, 0.0, 0)
        c.depth_to_u8(x0, u8, rows, cols)
        c.synchronize()
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 1
        assert_bit_equal(down(x1), d1, "level 1 of the spelt-out cascade")
        assert_bit_equal(down(x0), d0, "level 0 of the spelt-out cascade")
        assert np.array_equal(down(u8), want_u8)


def test_an_upload_into_the_coarsest_depth_image_makes_the_next_estimate_inject_again(oracle, lut):
    """the injection of the coarsest level (src/main.cpp:257-259) is only renewed when the annotation changed; rtdd_upload /
    rtdd_convert_to_float / rtdd_pyrup_depth INTO the coarsest RTDD_IMG_DEPTH through the library must count as such a change."""
    rows, cols = 270, 480
    bgr, ann = _bgr(rows, cols, 91)
    ref = Cascade(oracle, bgr, ann, lut, 1, threads=4)
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        levels = c.pyramid_create(rows, cols)
        c.pyramid_set_image(up(bgr)); c.pyramid_set_annotation(up(ann))
        ref.estimate(200); c.estimate_depth(200); c.synchronize()
        ptr, pitch, lr, lc = c.pyramid_image(rt.IMG_DEPTH, levels - 1)
        junk = np.full((lr, lc), 7.0, np.float32)
        c._check(rt.lib().rtdd_upload(c._h, C.c_void_p(ptr), C.c_size_t(pitch), C.c_void_p(junk.ctypes.data), C.c_size_t(lc * 4), C.c_size_t(lc * 4), C.c_int(lr)))
        ref.depth[levels - 1][...] = 7.0                       # what the reference's per-frame injection then restores at the labelled pixels
        ref.estimate(200); c.estimate_depth(200); c.synchronize()
        for l in range(levels):
            assert_bit_equal(c.pyramid_download(rt.IMG_DEPTH, l), ref.depth[l], f"level {l} after an upload into the coarsest depth image")


def test_a_healed_older_live_frame_runs_on_its_own_annotation(oracle, lut):
    """Two live frames in flight with DIFFERENT annotations (the second adds a stroke), the first frame's persistent launch times out:
    both are run again behind the caller's back.  Round 4's limit was that the healed older frame could see the newer frame's annotation;
    since round 5 every frame is replayed on the annotation pair it uploaded (include/rtdd.h): the older frame's map does not know the
    new stroke, the newer frame's map carries it, both honour their own labels, and the device ends up naming the newer frame's images.
    (The COARSE annotation levels only ever accumulate -- they may hold the new stroke during the older frame's replay: the documented
    remaining limit, which is why this test states properties and not the oracle's bits; with equal annotations the bits are the
    oracle's: test_live_frames_pipelined_match_the_oracle[True].)"""
    rows, cols = 540, 960
    bgr, ann = _bgr(rows, cols, 131)
    ref = Cascade(oracle, bgr, ann, lut, 1, threads=4)
    s1, e1 = ref.scribble[0].copy(), ref.edited[0].copy()
    s2, e2 = s1.copy(), e1.copy()
    free = np.argwhere(s1[100:400, 100:800] != 255)
    y0, x0 = (free[len(free) // 2] + 100).tolist()
    oracle.paint_image(x0, y0, 64, 21, e2, s2)                          # frame 2's new stroke (label 64)
    new = (s2 == 255) & (s1 != 255)
    assert new.sum() > 50
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        c.pyramid_create(rows, cols)
        c.pyramid_set_image(up(bgr)); c.pyramid_set_annotation(up(ann)); c.synchronize()
        h = [rt.host_image((rows, cols)), rt.host_image((rows, cols, 3)), rt.host_image((rows, cols)), rt.host_image((rows, cols, 3))]
        out = [rt.host_image((rows, cols)) for _ in range(2)]
        h[0].a[...] = s1; h[1].a[...] = e1; h[2].a[...] = s2; h[3].a[...] = e2
        c.set_option(rt.OPT_DEBUG_POLL_LIMIT_US, 3000); c.set_option(rt.OPT_DEBUG_WITHHOLD_TILE, 1)
        c.live_submit(h[0].a, h[1].a, out[0].a, 1000)
        c.live_submit(h[2].a, h[3].a, out[1].a, 1000)
        c.live_wait(); c.live_wait(); c.synchronize()
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 1
        m1, m2 = out[0].a.copy(), out[1].a.copy()
        assert np.array_equal(m1[s1 == 255], e1[..., 0][s1 == 255]) and np.array_equal(m2[s2 == 255], e2[..., 0][s2 == 255]), "each frame honours its own labels"
        assert (m2[new] == 64).all() and (m1[new] != 64).mean() > 0.9, "the older frame was replayed on ITS annotation: it does not know the new stroke"
        assert np.array_equal(c.pyramid_download(rt.IMG_SCRIBBLE, 0), s2) and np.array_equal(c.pyramid_download(rt.IMG_DEPTH_U8), m2)
        for x in h + out:
            x.free()


def test_no_device_memory_is_lost_over_context_lifetimes():
    """Forty contexts come and go -- single pyramids with pipelined live frames (page-locked and odd-pitch host images, both defocus paths)
    and batched pyramids, five sizes -- and the device's free memory afterwards is what it was before."""
    import torch
    probs = {}

    def lifetime(i):
        rows, cols = [(270, 480), (135, 241), (333, 517), (540, 960), (91, 90)][i % 5]
        if (rows, cols) not in probs:
            probs[(rows, cols)] = _bgr(rows, cols, 3)
        bgr, ann = probs[(rows, cols)]
        with rt.Context(0) as c:
            c.GPULoadWeights(0.4)
            dimg, dann = up(bgr), up(ann)
            if i % 3 == 0:
                c.pyramid_create_batch(rows, cols, 3)
                for b in range(3):
                    c.pyramid_select(b); c.pyramid_set_image(dimg); c.pyramid_set_annotation(dann)
                c.estimate_depth_batch(100)
            else:
                c.pyramid_create(rows, cols); c.pyramid_set_image(dimg); c.pyramid_set_annotation(dann)
                scr = rt.host_image((rows, cols)); ed = rt.host_image((rows, cols, 3)); out = rt.host_image((rows, cols))
                scr.a[...] = c.pyramid_download(rt.IMG_SCRIBBLE, 0); ed.a[...] = c.pyramid_download(rt.IMG_EDITED, 0)
                for _ in range(3):
                    c.live_submit(scr.a, ed.a, out.a, 100)
                while c.live_pending():
                    c.live_wait()
                d, dp, _, _ = c.pyramid_image(rt.IMG_DEPTH, 0); art = up(np.zeros_like(bgr))
                c.GPUSimulateDefocus(dimg, (d, dp), art, rows, cols)
                c.set_option(rt.OPT_DEFOCUS_PATH, 1); c.GPUSimulateDefocus(dimg, (d, dp), art, rows, cols)
                c.synchronize()
                for h in (scr, ed, out):
                    h.free()
            c.synchronize()

    def free_bytes():
        torch.cuda.synchronize(); torch.cuda.empty_cache()
        return torch.cuda.mem_get_info()[0]

    for i in range(6):
        lifetime(i)                                                      # (the runtime's own pools fill on first use)
    before = free_bytes()
    for i in range(40):
        lifetime(i)
    assert before - free_bytes() < (8 << 20)


@pytest.mark.parametrize("effect", ["defocus", "desaturation", "haze"])
@pytest.mark.parametrize("mode", ["pipelined", "one_at_a_time", "pipelined_withhold", "pageable"])
def test_live_frames_with_a_sticky_effect_match_the_oracle(oracle, lut, effect, mode):
    """The reference's frame WITH a sticky effect (src/main.cpp:190-230 next to :232-295; rtdd_live_submit_ex): every frame's u8 map is
    the oracle's n-th warm-started estimate AND every frame's artistic image is the oracle's effect on that estimate's depth map -- two
    frames in flight (the artistic image staged, downloaded by rtdd_live_wait), one frame at a time (its download queued on the compute
    stream), with the first frame's persistent launch timing out (both frames healed, effects rendered again), and into ordinary
    (pageable) host arrays."""
    rows, cols = 540, 960
    bgr, ann = _bgr(rows, cols, 77)
    ref = Cascade(oracle, bgr, ann, lut, 1, threads=oracle.max_threads())
    code = {"defocus": rt.EFFECT_DEFOCUS, "desaturation": rt.EFFECT_DESATURATION, "haze": rt.EFFECT_HAZE}[effect]

    def want_art(depth):
        if effect == "defocus":
            return oracle.defocus(bgr, depth, threads=oracle.max_threads())
        if effect == "desaturation":
            return oracle.desaturate(bgr, ref.gray[0], depth, 1)
        return oracle.haze(bgr, depth, 1)
    withhold = mode == "pipelined_withhold"
    depth_in_flight = 1 if mode == "one_at_a_time" else 2
    mk = _Pageable if mode == "pageable" else rt.host_image
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        c.pyramid_create(rows, cols)
        c.pyramid_set_image(up(bgr)); c.pyramid_set_annotation(up(ann)); c.synchronize()
        scr = rt.host_image((rows, cols)); ed = rt.host_image((rows, cols, 3))
        out = [mk((rows, cols)) for _ in range(2)]; art = [mk((rows, cols, 3)) for _ in range(2)]
        scr.a[...] = ref.scribble[0]; ed.a[...] = ref.edited[0]
        if withhold:
            c.set_option(rt.OPT_DEBUG_POLL_LIMIT_US, 3000); c.set_option(rt.OPT_DEBUG_WITHHOLD_TILE, 1)
        frames = 5
        got = []
        for n in range(frames):
            if n >= depth_in_flight:
                k = len(got); c.live_wait(); got.append((out[k % 2].a.copy(), art[k % 2].a.copy()))
            # (the middle frame without an effect: a frame's effect is its own, the slot's previous image must not leak into it)
            c.live_submit_ex(scr.a, ed.a, out[n % 2].a, code if n != 2 else rt.EFFECT_NONE, art[n % 2].a if n != 2 else None, 1000)
        while c.live_pending():
            k = len(got); c.live_wait(); got.append((out[k % 2].a.copy(), art[k % 2].a.copy()))
        c.synchronize()
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) == (1 if withhold else 0)
        for n in range(frames):
            ref.estimate(1000)
            assert np.array_equal(got[n][0], ref.depth_u8), f"frame {n}: u8 map"
            if n != 2:
                assert np.array_equal(got[n][1], want_art(ref.depth[0])), f"frame {n}: artistic image ({effect})"
        # RTDD_IMG_ARTISTIC on the device names the newest frame's artistic image
        assert np.array_equal(c.pyramid_download(rt.IMG_ARTISTIC), want_art(ref.depth[0]))
        with pytest.raises(rt.RtddError):
            c.live_submit_ex(scr.a, ed.a, out[0].a, code, None, 1000)        # an effect needs a host image
        with pytest.raises(rt.RtddError):
            c.live_submit_ex(scr.a, ed.a, out[0].a, 4, art[0].a, 1000)       # unknown effect


def test_a_retired_live_annotation_pointer_is_refused(oracle, lut):
    """After an uploading live frame the pyramid's level-0 scribble / edited images ARE the uploaded pair: a pointer rtdd_pyramid_image
    handed out before names a buffer no estimate reads.  Painting / uploading / converting through it would lose the strokes silently:
    RTDD_ERR_STATE instead; the pointers asked for again work, and the stroke painted through them reaches the next estimate."""
    rows, cols = 135, 241
    bgr, ann = _bgr(rows, cols, 5)
    ref = Cascade(oracle, bgr, ann, lut, 1, threads=4)
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        c.pyramid_create(rows, cols)
        c.pyramid_set_image(up(bgr)); c.pyramid_set_annotation(up(ann)); c.synchronize()
        old_s = c.pyramid_image(rt.IMG_SCRIBBLE, 0); old_e = c.pyramid_image(rt.IMG_EDITED, 0)
        c.GPUPaintImage(10, 10, 64, 3, (old_e[0], old_e[1]), (old_s[0], old_s[1]), rows, cols)     # fine: still the pyramid's
        oracle.paint_image(10, 10, 64, 3, ref.edited[0], ref.scribble[0])
        scr = rt.host_image((rows, cols)); ed = rt.host_image((rows, cols, 3)); out = rt.host_image((rows, cols))
        scr.a[...] = ref.scribble[0]; ed.a[...] = ref.edited[0]
        c.live_submit(scr.a, ed.a, out.a, 300); c.live_wait()
        ref.estimate(300)
        assert np.array_equal(out.a, ref.depth_u8)
        with pytest.raises(rt.RtddError) as e:
            c.GPUPaintImage(50, 60, 128, 4, (old_e[0], old_e[1]), (old_s[0], old_s[1]), rows, cols)
        assert e.value.status == 2 and "rtdd_pyramid_image again" in str(e.value)
        host = np.zeros((rows, cols), np.uint8)
        rc = rt.lib().rtdd_upload(c._h, C.c_void_p(old_s[0]), C.c_size_t(old_s[1]), C.c_void_p(host.ctypes.data), C.c_size_t(cols), C.c_size_t(cols), C.c_int(rows))
        assert rc == 2
        new_s = c.pyramid_image(rt.IMG_SCRIBBLE, 0); new_e = c.pyramid_image(rt.IMG_EDITED, 0)
        assert new_s[0] != old_s[0] and new_e[0] != old_e[0]
        c.GPUPaintImage(50, 60, 128, 4, (new_e[0], new_e[1]), (new_s[0], new_s[1]), rows, cols)
        oracle.paint_image(50, 60, 128, 4, ref.edited[0], ref.scribble[0])
        c.estimate_depth(300); c.synchronize()
        ref.estimate(300)
        assert_bit_equal(c.pyramid_download(rt.IMG_DEPTH, 0), ref.depth[0], "the stroke painted through the current pointers reached the estimate")
