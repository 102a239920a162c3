"""The harness's PNG reader/writer (zlib only; the dataset's annotations are PNG and the reference saves PNG) against Pillow.
Runs without a GPU: `rtdd_harness --convert` never touches the device."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "harness", "rtdd_harness")


def _read_pnm(path):
    with open(path, "rb") as f:
        magic = f.readline().strip(); w, h = map(int, f.readline().split()); f.readline()
        a = np.frombuffer(f.read(), np.uint8)
    return a.reshape(h, w, 3) if magic == b"P6" else a.reshape(h, w)


@pytest.fixture(scope="module")
def harness():
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "harness")])
    return BIN


@pytest.mark.parametrize("mode", ["L", "RGB", "RGBA", "P", "LA"])
def test_png_reader_matches_pillow(tmp_path, harness, mode):
    from PIL import Image
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:61, 0:83]
    base = np.stack([(xx * 3 + yy) % 256, (yy * 5) % 256, rng.integers(0, 256, xx.shape)], -1).astype(np.uint8)   # smooth + noisy: every filter type gets chosen
    im = Image.fromarray(base, "RGB")
    if mode == "P":
        im = im.quantize(37)
    elif mode != "RGB":
        im = im.convert(mode)
    im.save(tmp_path / "in.png", optimize=True)
    want = np.array(im.convert("L" if mode in ("L", "LA") else "RGB"))
    out = tmp_path / ("out.pgm" if want.ndim == 2 else "out.ppm")
    subprocess.check_call([harness, "--convert", str(tmp_path / "in.png"), str(out)])
    assert np.array_equal(_read_pnm(out), want)


@pytest.mark.parametrize("ch", [1, 3])
def test_png_writer_is_read_back_by_pillow(tmp_path, harness, ch):
    from PIL import Image
    rng = np.random.default_rng(11)
    a = rng.integers(0, 256, (45, 70) if ch == 1 else (45, 70, 3), dtype=np.uint8)
    with open(tmp_path / "in.pnm", "wb") as f:
        f.write(b"%s\n70 45\n255\n" % (b"P5" if ch == 1 else b"P6")); f.write(a.tobytes())
    subprocess.check_call([harness, "--convert", str(tmp_path / "in.pnm"), str(tmp_path / "out.png")])
    got = Image.open(tmp_path / "out.png")
    assert got.mode == ("L" if ch == 1 else "RGB") and np.array_equal(np.array(got), a)


def test_dataset_annotation_rule_on_png(tmp_path, harness):
    """An RGB annotation PNG with R = G = B (how the bundled annotations are stored) decodes to that value, as cv::imread(path, 0) does."""
    from PIL import Image
    v = np.random.default_rng(3).choice(np.array([0, 32, 64, 128, 192, 254], np.uint8), (20, 30))
    Image.fromarray(np.repeat(v[..., None], 3, 2), "RGB").save(tmp_path / "ann.png")
    subprocess.check_call([harness, "--convert", str(tmp_path / "ann.png"), str(tmp_path / "ann.ppm")])
    assert np.array_equal(_read_pnm(tmp_path / "ann.ppm")[..., 0], v)
