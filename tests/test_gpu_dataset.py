"""north_star: "reproduce the reference CUDA path's depth map on the bundled dataset" -- all twelve dataset pairs at their
own resolution (-m gpu), the whole call sequence of src/main.cpp:93-113,160-173,232-295 and the three effects
(:190-230), GPU against the oracle run on this box, every pyramid level bit for bit.

Nine of the twelve have odd dimensions somewhere in their pyramid (1280x853, 1280x841, 1280x685, 910x910 -> 455 ...): those
take the host `cv::pyrUp`-with-explicit-size branch of src/main.cpp:272-279 and the ceil/floor gray-pyramid quirk of
SURVEY A.6 (gray level sizes follow a ceil chain, the solver reads their floor-sized window) at several levels; the even
ones (Dog, Pigs) take the `cv::cuda::pyrUp` branch throughout.  Stated tolerance 1e-4 (BASELINE north_star); asserted:
bit-exact for the solver, the annotation passes and all three effects (haze too since round 3: the same deterministic exp on both sides)."""
import os
import subprocess

import numpy as np
import pytest

import realtimedepthdiffusion_amd as rt
from cascade_ref import Cascade
from dataset_util import MANIFEST, PAIRS, annotation_path, image_path, load_pair
from golden_util import sha
from gpu_util import assert_bit_equal, down, up

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "harness", "rtdd_harness")


def test_the_dataset_covers_both_pyrup_branches_and_the_odd_size_quirk():
    odd = [n for n in PAIRS if MANIFEST[n]["sizes"] != MANIFEST[n]["gray_sizes"]]
    even = [n for n in PAIRS if n not in odd]
    assert len(PAIRS) == 12 and len(odd) >= 9 and set(even) >= {"Dog", "Pigs"}


@pytest.mark.parametrize("name", PAIRS)
def test_dataset_pair_whole_estimate_and_effects(oracle, lut, name):
    bgr, ann, e = load_pair(name)
    rows, cols = bgr.shape[:2]
    threads = min(8, oracle.max_threads())
    ref = Cascade(oracle, bgr, ann, lut, 1, threads=threads)
    ref.estimate(1000)
    # the oracle on this box == the oracle in the authoring container (libm expf of the LUT, OpenMP row split)
    assert ref.P == e["levels"] and [list(s) for s in ref.size] == e["sizes"]
    assert [sha(ref.depth[l]) for l in range(ref.P)] == e["depth_sha_c1"] and sha(ref.depth_u8) == e["depth_u8_sha_c1"]
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        assert c.pyramid_create(rows, cols) == ref.P
        c.pyramid_set_image(up(bgr)); c.pyramid_set_annotation(up(ann))
        c.estimate_depth(1000); c.synchronize()
        for l in range(ref.P):
            r, k = ref.size[l]
            got_gray = c.pyramid_download(rt.IMG_GRAY, l)
            assert list(got_gray.shape) == e["gray_sizes"][l] and np.array_equal(got_gray, ref.gray[l]), f"{name} gray {l}"
            assert np.array_equal(c.pyramid_download(rt.IMG_SCRIBBLE, l), ref.scribble[l]), f"{name} scribble {l}"
            assert np.array_equal(c.pyramid_download(rt.IMG_EDITED, l)[..., 0], ref.edited[l][..., 0]), f"{name} edited {l}"
        for l in range(ref.P - 1, -1, -1):
            got = c.pyramid_download(rt.IMG_DEPTH, l)
            assert np.abs(got - ref.depth[l]).max() <= 1e-4
            assert_bit_equal(got, ref.depth[l], f"{name} depth level {l}")
        assert np.array_equal(c.pyramid_download(rt.IMG_DEPTH_U8), ref.depth_u8)
        # the three effects on the finest depth (src/main.cpp:190-230), through the reference-named entry points
        o = up(bgr); d = up(ref.depth[0]); gray = up(ref.gray[0]); art = up(np.zeros_like(bgr))
        c.GPUSimulateDesaturation(o, gray, d, art, rows, cols)
        want = oracle.desaturate(bgr, ref.gray[0], ref.depth[0], 1)
        assert sha(want) == e["desaturate_sha"] and np.array_equal(down(art), want), f"{name} desaturation"
        c.GPUSimulateDefocus(o, d, art, rows, cols)
        want = oracle.defocus(bgr, ref.depth[0], threads=oracle.max_threads())
        assert sha(want) == e["defocus_sha"] and np.array_equal(down(art), want), f"{name} defocus"
        c.GPUSimulateHaze(o, d, art, rows, cols)
        want = oracle.haze(bgr, ref.depth[0], 1)
        assert sha(want) == e["haze_sha"]
        assert np.array_equal(down(art), want), f"{name} haze"
        # the non-contracted variant (nvcc -fmad=false), against the hashes recorded with the fixture
        c.set_option(rt.OPT_FP_CONTRACT, 0)
        c.pyramid_set_image(up(bgr)); c.pyramid_set_annotation(up(ann))        # a new image: cold start (main.cpp:136)
        c.estimate_depth(1000); c.synchronize()
        assert [sha(c.pyramid_download(rt.IMG_DEPTH, l)) for l in range(ref.P)] == e["depth_sha_c0"], f"{name} uncontracted"
        assert sha(c.pyramid_download(rt.IMG_DEPTH_U8)) == e["depth_u8_sha_c0"]


@pytest.mark.parametrize("name,ext", [("Flower", "png"), ("Arara", "png"), ("Straw", "png"), ("Dog", "jpg"), ("Heidelberg", "jpg"), ("WomanParasol", "jpg")])
def test_harness_on_the_dataset_files(oracle, lut, tmp_path, name, ext):
    """The C++ harness (host code over the C ABI) fed the dataset's files as they are: the JPEG the reference ships (sequential and
    progressive ones, through the harness's own decoder) or its lossless PNG copy, PNG annotation, PNG out."""
    from PIL import Image
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "harness")])
    bgr, ann, e = load_pair(name)
    out = subprocess.check_output([BIN, "-i", image_path(name)[:-3] + ext, "-a", annotation_path(name), "-o", str(tmp_path) + "/", "--effect", "defocus", "--png"], text=True)
    assert "Saving images" in out
    depth_u8 = np.array(Image.open(tmp_path / "DepthMap.png"))
    assert sha(depth_u8) == e["depth_u8_sha_c1"]
    assert sha(np.ascontiguousarray(np.array(Image.open(tmp_path / "ArtisticEffect.png"))[..., ::-1])) == e["defocus_sha"]
