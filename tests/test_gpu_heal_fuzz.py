"""Randomised sequences through the self-healing path (-m gpu): solves chained in place on several images, whole estimates, paints,
weight-table changes and synchronising calls of every kind, with a hand-off flag withheld from a random point on so that some
persistent launch in the middle of a queue times out.  Whatever the sequence, every image the caller sees after a synchronising
call must be what the same sequence gives on the CPU oracle (src/GPUSolver.cu:311-314: the solver always leaves a valid depth map),
the context must have healed at most once, and nothing may hang."""
import ctypes as C

import numpy as np
import pytest

import realtimedepthdiffusion_amd as rt
from cascade_ref import Cascade
from gpu_util import assert_bit_equal, down, up
from realtimedepthdiffusion_amd.synth import make_problem
from test_gpu_cascade import _bgr

pytestmark = pytest.mark.gpu
ROWS, COLS = 270, 480          # the smallest level the cost model runs persistently (tile 9: 40 workgroups)


@pytest.mark.parametrize("seed", range(6))
def test_random_sequences_of_solves_heal_to_the_oracles_bits(oracle, seed):
    rng = np.random.default_rng(100 + seed)
    probs = [make_problem(ROWS, COLS, seed=500 + seed * 3 + i) for i in range(3)]
    luts = {0.4: oracle.load_weights(0.4), 0.25: oracle.load_weights(0.25)}
    beta = 0.4
    host = [p["depth"].copy() for p in probs]                 # what the oracle says each image holds
    with rt.Context(0) as c:
        c.GPUAllocateDeviceMemory(ROWS, COLS, 1); c.GPULoadWeights(beta)
        dev = [up(p["depth"]) for p in probs]; masks = [up(p["mask"]) for p in probs]; grays = [up(p["gray"]) for p in probs]
        armed_at = int(rng.integers(0, 6))
        for step in range(14):
            if step == armed_at:
                c.set_option(rt.OPT_DEBUG_POLL_LIMIT_US, 2000); c.set_option(rt.OPT_DEBUG_WITHHOLD_TILE, int(rng.integers(1, 30)))
            op = rng.choice(["solve", "solve", "solve", "sync", "download", "weights"])
            if op == "solve":
                i = int(rng.integers(0, 3)); n = int(rng.choice([24, 40, 64, 100]))
                c.GPUMatrixFreeSolver(dev[i], masks[i], grays[i], ROWS, COLS, beta, n, 0.0, 0)
                host[i] = oracle.solve(host[i], probs[i]["mask"], probs[i]["gray"], n, 0, 0, luts[beta], 1, threads=4)
            elif op == "sync":
                c.synchronize()
                for i in range(3):
                    assert_bit_equal(down(dev[i]), host[i], f"seed {seed} step {step}: image {i} after rtdd_ctx_synchronize")
            elif op == "download":
                i = int(rng.integers(0, 3))
                out = np.empty((ROWS, COLS), np.float32)
                c._check(rt.lib().rtdd_download(c._h, C.c_void_p(out.ctypes.data), C.c_size_t(COLS * 4), C.c_void_p(dev[i].data_ptr()),
                                                C.c_size_t(dev[i].stride(0) * 4), C.c_size_t(COLS * 4), C.c_int(ROWS)))
                assert_bit_equal(out, host[i], f"seed {seed} step {step}: image {i} through rtdd_download")
            else:
                beta = 0.25 if beta == 0.4 else 0.4
                c.GPULoadWeights(beta)                         # settles the log: solves made with the old table are healed with it
        c.synchronize()
        for i in range(3):
            assert_bit_equal(down(dev[i]), host[i], f"seed {seed}: image {i} at the end")
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) <= 1
        _HEALS.append(c.get_option(rt.OPT_TIMEOUT_HEALS))


_HEALS = []


def test_the_sequences_above_did_heal():
    """(the fuzz is only a test of the healing path if time-outs really happened in it)"""
    assert len(_HEALS) == 6 and sum(_HEALS) >= 4, _HEALS


@pytest.mark.parametrize("seed", range(3))
def test_random_sequences_of_estimates_and_paints_heal(oracle, lut, seed):
    rng = np.random.default_rng(300 + seed)
    rows, cols = 540, 960
    bgr, ann = _bgr(rows, cols, 40 + seed)
    ref = Cascade(oracle, bgr, ann, lut, 1, threads=oracle.max_threads())
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4)
        levels = c.pyramid_create(rows, cols)
        c.pyramid_set_image(up(bgr)); c.pyramid_set_annotation(up(ann))
        sp, spitch, _, _ = c.pyramid_image(rt.IMG_SCRIBBLE, 0); ep, epitch, _, _ = c.pyramid_image(rt.IMG_EDITED, 0)
        armed_at = int(rng.integers(0, 4))
        for step in range(7):
            if step == armed_at:
                c.set_option(rt.OPT_DEBUG_POLL_LIMIT_US, 2000); c.set_option(rt.OPT_DEBUG_WITHHOLD_TILE, int(rng.integers(1, 20)))
            if rng.random() < 0.4:
                x, y, lab, rad = int(rng.integers(0, cols)), int(rng.integers(0, rows)), int(rng.choice([0, 64, 128, 192, 254])), int(rng.integers(5, 30))
                oracle.paint_image(x, y, lab, rad, ref.edited[0], ref.scribble[0])
                c.GPUPaintImage(x, y, lab, rad, (ep, epitch), (sp, spitch), rows, cols)
            iters = int(rng.choice([200, 400]))
            ref.estimate(iters); c.estimate_depth(iters)
            c.synchronize()                                     # (a paint between two UNSYNCHRONISED estimates is the documented limit of the replay)
            for l in range(levels):
                assert_bit_equal(c.pyramid_download(rt.IMG_DEPTH, l), ref.depth[l], f"seed {seed} step {step}: level {l}")
            assert np.array_equal(c.pyramid_download(rt.IMG_DEPTH_U8), ref.depth_u8)
        assert c.get_option(rt.OPT_TIMEOUT_HEALS) <= 1
