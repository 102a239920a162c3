"""Batched estimates (rtdd_pyramid_create_batch / rtdd_estimate_depth_batch: BASELINE configs[3], independent images on one GPU, every
pyramid level of all images in the same launches -- blockIdx.z = image).  The bar: every image's maps, level by level, are bit for bit
what a single-image pyramid gives for that image (and what the oracle's cascade gives), warm-started estimates included, and the
self-healing path covers a batch like a single estimate.  (-m gpu)"""
import numpy as np
import pytest

import realtimedepthdiffusion_amd as rt
from cascade_ref import Cascade
from gpu_util import assert_bit_equal, up
from test_gpu_cascade import _bgr

pytestmark = pytest.mark.gpu


def _single(rows, cols, bgr, ann, iters, estimates=2, paint=None):
    """[estimate][level] depth images and [estimate] u8 maps of ONE image on a single-image pyramid."""
    out = []
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        levels = c.pyramid_create(rows, cols)
        c.pyramid_set_image(up(bgr)); c.pyramid_set_annotation(up(ann))
        for e in range(estimates):
            if paint is not None and e == 1:
                sp, spitch, _, _ = c.pyramid_image(rt.IMG_SCRIBBLE, 0); ep, epitch, _, _ = c.pyramid_image(rt.IMG_EDITED, 0)
                c.GPUPaintImage(*paint, (ep, epitch), (sp, spitch), rows, cols)
            c.estimate_depth(iters); c.synchronize()
            out.append(([c.pyramid_download(rt.IMG_DEPTH, l) for l in range(levels)], c.pyramid_download(rt.IMG_DEPTH_U8)))
    return out


def _compare(c, b, want, what):
    c.pyramid_select(b)
    depth, u8 = want
    for l, d in enumerate(depth):
        assert_bit_equal(c.pyramid_download(rt.IMG_DEPTH, l), d, f"{what}: image {b}, level {l}")
    assert np.array_equal(c.pyramid_download(rt.IMG_DEPTH_U8), u8), f"{what}: image {b}, u8 map"


@pytest.mark.parametrize("rows,cols,images,iters", [(270, 480, 5, 400), (135, 241, 3, 300), (90, 91, 7, 100), (1080, 1920, 3, 1000), (540, 960, 9, 600)])
def test_batched_estimates_equal_single_image_estimates(rows, cols, images, iters):
    """Different images and annotations per slot; the cold estimate and the warm-started one behind it; a stroke painted into ONE
    image of the batch between the two (the annotation flag is one per batch: the other images must come out unchanged by that)."""
    data = [_bgr(rows, cols, 200 + 7 * b) for b in range(images)]
    paint = (cols // 3, rows // 2, 192, max(6, rows // 20))
    want = [_single(rows, cols, bgr, ann, iters, paint=paint if b == 1 else None) for b, (bgr, ann) in enumerate(data)]
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        c.pyramid_create_batch(rows, cols, images)
        assert rt.lib().rtdd_pyramid_batch(c._h) == images
        for b, (bgr, ann) in enumerate(data):
            c.pyramid_select(b)
            c.pyramid_set_image(up(bgr)); c.pyramid_set_annotation(up(ann))
        c.estimate_depth_batch(iters); c.synchronize()
        for b in range(images):
            _compare(c, b, want[b][0], f"{cols}x{rows} x {images}, cold")
        c.pyramid_select(1)
        sp, spitch, _, _ = c.pyramid_image(rt.IMG_SCRIBBLE, 0); ep, epitch, _, _ = c.pyramid_image(rt.IMG_EDITED, 0)
        c.GPUPaintImage(*paint, (ep, epitch), (sp, spitch), rows, cols)
        c.estimate_depth_batch(iters); c.synchronize()
        for b in range(images):
            _compare(c, b, want[b][1], f"{cols}x{rows} x {images}, warm start")
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 0


def test_a_batch_matches_the_oracles_cascade(oracle, lut):
    """... and against the CPU restatement directly (not only against the library's own single-image path)."""
    rows, cols, images = 256, 256, 4
    data = [_bgr(rows, cols, 300 + b) for b in range(images)]
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        levels = c.pyramid_create_batch(rows, cols, images)
        for b, (bgr, ann) in enumerate(data):
            c.pyramid_select(b); c.pyramid_set_image(up(bgr)); c.pyramid_set_annotation(up(ann))
        c.estimate_depth_batch(1000); c.synchronize()
        for b, (bgr, ann) in enumerate(data):
            ref = Cascade(oracle, bgr, ann, lut, 1, threads=oracle.max_threads())
            ref.estimate(1000)
            c.pyramid_select(b)
            for l in range(levels):
                assert_bit_equal(c.pyramid_download(rt.IMG_DEPTH, l), ref.depth[l], f"image {b} level {l} against the oracle")
            assert np.array_equal(c.pyramid_download(rt.IMG_DEPTH_U8), ref.depth_u8)


def test_single_image_calls_address_the_selected_image_of_a_batch():
    """rtdd_estimate_depth and rtdd_refine_depth on a batched pyramid run the selected image only; the others keep their state."""
    rows, cols, images, iters = 270, 480, 3, 300
    data = [_bgr(rows, cols, 400 + b) for b in range(images)]
    want = [_single(rows, cols, bgr, ann, iters, estimates=1) for bgr, ann in data]
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        levels = c.pyramid_create_batch(rows, cols, images)
        for b, (bgr, ann) in enumerate(data):
            c.pyramid_select(b); c.pyramid_set_image(up(bgr)); c.pyramid_set_annotation(up(ann))
        c.pyramid_select(2)
        c.estimate_depth(iters); c.synchronize()
        _compare(c, 2, want[2][0], "the selected image alone")
        c.pyramid_select(0)
        assert np.all(c.pyramid_download(rt.IMG_DEPTH, 0) == 255.0), "an image nobody estimated keeps its initial depth (src/main.cpp:136)"
        c.estimate_depth_batch(iters); c.synchronize()          # image 2 is warm-started now, 0 and 1 are cold
        _compare(c, 0, want[0][0], "cold image of a mixed batch"); _compare(c, 1, want[1][0], "cold image of a mixed batch")
        with pytest.raises(rt.RtddError):
            c.pyramid_select(images)


def test_a_batch_heals_a_timed_out_persistent_level(capfd):
    """One tile's hand-off flag withheld in every image of the batch: the first persistent level times out, every copy-back behind it
    stores nothing, the synchronising call runs the whole batch again from that level without persistence: all maps as if nothing had
    happened."""
    rows, cols, images, iters = 1080, 1920, 2, 1000
    data = [_bgr(rows, cols, 500 + b) for b in range(images)]
    want = [_single(rows, cols, bgr, ann, iters, estimates=1) for bgr, ann in data]
    capfd.readouterr()
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        c.pyramid_create_batch(rows, cols, images)
        for b, (bgr, ann) in enumerate(data):
            c.pyramid_select(b); c.pyramid_set_image(up(bgr)); c.pyramid_set_annotation(up(ann))
        c.set_option(rt.OPT_DEBUG_POLL_LIMIT_US, 3000); c.set_option(rt.OPT_DEBUG_WITHHOLD_TILE, 1)
        c.estimate_depth_batch(iters); c.synchronize()
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 1
        assert capfd.readouterr().err.count("rtdd: persistent sweep kernel") == 1
        for b in range(images):
            _compare(c, b, want[b][0], "healed batch")


# ---- the batches bench.py TIMES (batch64_1080p_estimate: 64 images at N = 1, 8 per GPU at N = 8) --------------------------------------
# The per-level tile / depth / persistence choice is a function of the batch size (sweep_blocked.hip config_cost / launch_sweeps_blocked),
# so the batch sizes of the other tests (2 .. 9) exercise other kernel configurations than the ones the bench line's numbers come from.
# Pinned here: what each level of those two batches runs -- (tile, sweeps per launch or exchange, persistent, images per sweep launch),
# level 0 = 1920 x 1080 -- so that a change of the cost model shows up as a failing test next to the figure it would change.
BENCH_BATCH_CHOICES = {
    # 120 x 67: a workgroup per image, all 1000 sweeps in one launch; 240 x 135: four persistent tiles per image; 480 x 270 and 960 x 540: one launch
    # per 8 sweeps over all images; 1920 x 1080: image after image, each the persistent launch a single solve gets (profiles/r05_batch_level_ab.txt)
    64: {4: (4, 1000, 0, 64), 3: (4, 4, 1, 64), 2: (6, 8, 0, 64), 1: (6, 8, 0, 64), 0: (4, 8, 1, 1)},
    8: {4: (4, 1000, 0, 8), 3: (9, 12, 1, 8), 2: (8, 8, 1, 8), 1: (6, 12, 0, 8), 0: (4, 8, 1, 1)},
}


def bench_batch_problem(b, rows=1080, cols=1920):
    """Image b of bench.py's batch (batch_estimates / the batch64_1080p_estimate workload: seeds 1234 + b)."""
    from realtimedepthdiffusion_amd.synth import make_problem
    p = make_problem(rows, cols, seed=1234 + b)
    return np.repeat(p["gray"][..., None], 3, 2), np.where(p["mask"] == 255, p["edited"][..., 0], 32).astype(np.uint8)


@pytest.mark.parametrize("images", [64, 8])
def test_the_batches_the_bench_times_match_the_oracle(images, oracle, lut):
    """bench.py's 64 x 1080p batch (and the 8 images a rank owns at N = 8), cold and warm-started: the per-level choices are the pinned
    ones, and every level + the u8 map of the sampled images equals the oracle's cascade bit for bit (src/main.cpp:232-295)."""
    rows, cols = 1080, 1920
    sample = [b for b in (0, 1, 7, 8, 31, 32, 62, 63) if b < images] if images > 8 else list(range(images))
    refs = {}
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        levels = c.pyramid_create_batch(rows, cols, images)
        for b in range(images):
            bgr, ann = bench_batch_problem(b)
            c.pyramid_select(b); c.pyramid_set_image(up(bgr)); c.pyramid_set_annotation(up(ann))
            if b in sample:
                refs[b] = Cascade(oracle, bgr, ann, lut, 1, threads=oracle.max_threads())
        for what in ("cold", "warm start"):
            c.estimate_depth_batch(1000); c.synchronize()
            got = {l: (i.tile, i.temporal_depth, i.persistent, n) for l in range(levels) for i, n in [c.pyramid_level_info(l)]}
            assert got == BENCH_BATCH_CHOICES[images], f"{images} x 1080p, {what}: the per-level kernel choices changed: {got}"
            assert all(c.pyramid_level_info(l)[0].kernel == 2 for l in range(levels))
            for b in sample:
                refs[b].estimate(1000)
                _compare(c, b, (refs[b].depth, refs[b].depth_u8), f"{images} x 1080p as bench.py times it, {what}, against the oracle")
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 0
