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
