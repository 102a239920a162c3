"""Helpers shared by the -m gpu parity tests."""
import numpy as np

import realtimedepthdiffusion_amd as rt


def up(a, align=512):
    return rt.device_image(a, "cuda:0", align)


def down(t):
    return rt.to_host(t)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def assert_bit_equal(got, want, what=""):
    got = np.ascontiguousarray(got); want = np.ascontiguousarray(want)
    if not np.array_equal(bits(got), bits(want)):
        d = np.abs(got.astype(np.float64) - want.astype(np.float64))
        n = int((bits(got) != bits(want)).sum())
        raise AssertionError(f"{what}: {n} of {got.size} values differ, max abs diff {np.nanmax(d):.3e}")
