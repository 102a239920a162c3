"""Seeded synthetic inputs for the solver benchmarks and full-size parity tests.

Follows SURVEY.md section 8(d): gray = 3 octaves of bilinear value noise + random rectangles;
mask = ~10 % Dirichlet coverage from 24 square-brush polyline strokes (brush = 2 % of the
short side, cf. /root/reference/src/main.cpp:154), one label of {0,64,128,192,254} per stroke
(src/main.cpp:41-42); initial depth 255 with labels injected (src/main.cpp:136,257).
numpy only -- no torch, no oracle, no GPU.
"""
import numpy as np

LABELS = np.array([0, 64, 128, 192, 254], np.uint8)


def _value_noise(rng, rows, cols, cell):
    gh, gw = rows // cell + 2, cols // cell + 2
    g = rng.random((gh, gw), dtype=np.float32)
    y = np.arange(rows, dtype=np.float32) / cell
    x = np.arange(cols, dtype=np.float32) / cell
    y0 = y.astype(np.int32); x0 = x.astype(np.int32)
    fy = (y - y0)[:, None]; fx = (x - x0)[None, :]
    top = g[y0][:, x0] * (1 - fx) + g[y0][:, x0 + 1] * fx
    bot = g[y0 + 1][:, x0] * (1 - fx) + g[y0 + 1][:, x0 + 1] * fx
    return top * (1 - fy) + bot * fy


def make_problem(rows, cols, seed=1234, coverage=0.10, strokes=24):
    """Returns dict(gray u8 [r,c], mask u8 [r,c] (255 = Dirichlet, else 32), edited u8 [r,c,3],
    depth f32 [r,c])."""
    rng = np.random.default_rng(seed)
    g = 110.0 * _value_noise(rng, rows, cols, 64) + 50.0 * _value_noise(rng, rows, cols, 16) \
        + 6.0 * _value_noise(rng, rows, cols, 4) + 30.0
    for _ in range(int(rng.integers(4, 9))):
        y0 = int(rng.integers(0, rows)); x0 = int(rng.integers(0, cols))
        h = int(rng.integers(max(rows // 16, 1), max(rows // 3, 2))); w = int(rng.integers(max(cols // 16, 1), max(cols // 3, 2)))
        g[y0:y0 + h, x0:x0 + w] += float(rng.integers(-80, 81))
    gray = np.ascontiguousarray(np.clip(g, 0, 255).astype(np.uint8))

    mask = np.full((rows, cols), 32, np.uint8)
    label = np.zeros((rows, cols), np.uint8)
    brush = max(int(min(rows, cols) * 0.02), 1)
    half = brush // 2
    total_len = coverage * rows * cols / (strokes * (2 * half + 1))
    for _ in range(strokes):
        lab = LABELS[int(rng.integers(0, len(LABELS)))]
        nseg = int(rng.integers(2, 5))
        px = float(rng.uniform(0, cols)); py = float(rng.uniform(0, rows))
        for _s in range(nseg):
            ang = float(rng.uniform(0, 2 * np.pi)); seg = total_len / nseg
            nstamp = max(int(seg / max(half, 1)), 1)
            dx = np.cos(ang) * seg / nstamp; dy = np.sin(ang) * seg / nstamp
            for _k in range(nstamp):
                cx = int(px); cy = int(py)
                ya, yb = max(cy - half, 0), min(cy + half + 1, rows)
                xa, xb = max(cx - half, 0), min(cx + half + 1, cols)
                if ya < yb and xa < xb:
                    mask[ya:yb, xa:xb] = 255
                    label[ya:yb, xa:xb] = lab
                px = min(max(px + dx, 0), cols - 1); py = min(max(py + dy, 0), rows - 1)
    edited = np.zeros((rows, cols, 3), np.uint8)
    edited[mask == 255] = label[mask == 255][:, None]
    depth = np.full((rows, cols), 255.0, np.float32)
    depth[mask == 255] = label[mask == 255].astype(np.float32)
    return {"gray": gray, "mask": mask, "edited": edited, "depth": depth}


def pitched(a, align=512):
    """Copy of `a` whose rows are padded to `align` bytes (mimics cudaMallocPitch); returns
    the padded base array and the logical view."""
    rows = a.shape[0]
    row_bytes = a[0].nbytes
    pitch = (row_bytes + align - 1) // align * align
    base = np.zeros((rows, pitch), np.uint8)
    view = base[:, :row_bytes].view(a.dtype).reshape(a.shape)
    view[...] = a
    return base, view
