"""Independent O(N) restatement of GPUSimulateDefocus (src/GPUDepthEffect.cu:29-72) for the full-size tests: a 64-bit
summed-area table in numpy.  Test infrastructure; pinned against the oracle's literal gather in tests/test_oracle.py."""
import numpy as np


def effect_inputs(rows, cols, seed):
    rng = np.random.default_rng(seed)
    orig = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    # depth with every regime: large flat near / far regions, a ramp, noise; a few out-of-range and exact-integer values
    yy, xx = np.mgrid[0:rows, 0:cols]
    depth = (255.0 * xx / cols).astype(np.float32)
    depth[: rows // 4] = 255.0
    depth[rows // 4: rows // 3] = 0.0
    depth[rows // 2:] += rng.uniform(-40, 40, (rows - rows // 2, cols)).astype(np.float32)
    depth[::97, ::89] = np.float32(300.0); depth[5::101, 3::83] = np.float32(-7.5)
    return orig, np.ascontiguousarray(depth)


def defocus_by_summed_area_table(orig, depth):
    """GPUSimulateDefocus restated in O(N) with exact 64-bit integer sums (numpy): window [y - k/2, y + k/2) x [x - k/2, x + k/2)
    clipped to the image, k = (int)((double)((float)K * depth) / 255.0) with K = (int)(0.025 * sqrtf(rows^2 + cols^2)),
    out = (uchar)(sum / count) in f32, count == 0 -> the original pixel; out-of-range depth as the oracle defines it."""
    rows, cols = depth.shape
    K = int(np.float32(0.025) * 0 + 0.025 * float(np.sqrt(np.float32(rows * rows + cols * cols))))
    kf = (np.float32(K) * depth).astype(np.float64) / 255.0
    k = np.trunc(kf).astype(np.int64)
    h = np.where(k >= 0, k // 2, -((-k) // 2))                     # C integer division truncates toward zero
    y = np.arange(rows)[:, None]; x = np.arange(cols)[None, :]
    y0 = np.clip(y - h, 0, rows); y1 = np.clip(y + h, 0, rows); x0 = np.clip(x - h, 0, cols); x1 = np.clip(x + h, 0, cols)
    cnt = np.maximum(y1 - y0, 0) * np.maximum(x1 - x0, 0)
    out = np.empty_like(orig)
    for c in range(3):
        S = np.zeros((rows + 1, cols + 1), np.int64)
        np.cumsum(np.cumsum(orig[..., c].astype(np.int64), 0), 1, out=S[1:, 1:])
        s = S[y1, x1] - S[y0, x1] - S[y1, x0] + S[y0, x0]
        with np.errstate(divide="ignore", invalid="ignore"):
            q = s.astype(np.float32) / cnt.astype(np.float32)
        out[..., c] = np.where(cnt > 0, np.clip(np.trunc(q), 0, 255), orig[..., c]).astype(np.uint8)
    return out
