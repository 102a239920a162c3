"""The C++ headless harness (harness/rtdd_harness, host code over the C ABI) end to end on a golden crop (-m gpu)."""
import os
import subprocess

import numpy as np
import pytest

from golden_util import NAMES, load

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "harness", "rtdd_harness")


def _write_pnm(path, a):
    with open(path, "wb") as f:
        f.write(b"%s\n%d %d\n255\n" % (b"P6" if a.ndim == 3 else b"P5", a.shape[1], a.shape[0]))
        f.write(np.ascontiguousarray(a).tobytes())


def _read_pnm(path):
    with open(path, "rb") as f:
        magic = f.readline().strip(); w, h = map(int, f.readline().split()); f.readline()
        a = np.frombuffer(f.read(), np.uint8)
    return a.reshape(h, w, 3) if magic == b"P6" else a.reshape(h, w)


@pytest.mark.parametrize("name", NAMES[:2])
@pytest.mark.parametrize("effect,key", [("desaturation", "desaturate_c1"), ("defocus", "defocus")])
def test_harness_reproduces_golden(tmp_path, name, effect, key):
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "harness")])
    g = load(name)
    _write_pnm(tmp_path / "img.ppm", g["bgr"][..., ::-1])             # files are RGB; the harness converts like cv::imread
    _write_pnm(tmp_path / "ann.pgm", g["annotation"])
    out = subprocess.check_output([BIN, "-i", str(tmp_path / "img.ppm"), "-a", str(tmp_path / "ann.pgm"), "-o", str(tmp_path) + "/",
                                   "--effect", effect], text=True)
    assert "Processing Time" in out and "Saving images" in out
    assert np.array_equal(_read_pnm(tmp_path / "DepthMap.pgm"), g["depth_u8"])
    assert np.array_equal(_read_pnm(tmp_path / "ArtisticEffect.ppm"), g[key][..., ::-1])


def test_harness_png_in_png_out(tmp_path):
    """The dataset's own formats: RGB annotation PNG (R = G = B), PNG image, --png outputs as the reference saves them."""
    from PIL import Image
    g = load(NAMES[1])
    Image.fromarray(np.ascontiguousarray(g["bgr"][..., ::-1]), "RGB").save(tmp_path / "img.png")
    Image.fromarray(np.repeat(g["annotation"][..., None], 3, 2), "RGB").save(tmp_path / "ann.png")
    out = subprocess.check_output([BIN, "-i", str(tmp_path / "img.png"), "-a", str(tmp_path / "ann.png"), "-o", str(tmp_path) + "/", "--effect", "defocus", "--png"], text=True)
    assert "Saving images" in out
    assert np.array_equal(np.array(Image.open(tmp_path / "DepthMap.png")), g["depth_u8"])
    assert np.array_equal(np.array(Image.open(tmp_path / "ArtisticEffect.png")), g["defocus"][..., ::-1])


def test_harness_batch_and_paint(tmp_path):
    g = load(NAMES[0])
    _write_pnm(tmp_path / "img.ppm", g["bgr"][..., ::-1])
    # no annotation file: two brush samples instead (mouse drag, main.cpp:46-62); batch of 3 on 1 device
    out = subprocess.check_output([BIN, "-i", str(tmp_path / "img.ppm"), "-o", str(tmp_path) + "/", "--paint", "40,40,0,9", "--paint", "200,180,254,9",
                                   "--batch", "3", "--devices", "1", "--iters", "200"], text=True)
    assert "3 estimate(s) on 1 device(s)" in out
    d = _read_pnm(tmp_path / "DepthMap.pgm")
    assert d[40, 40] == 0 and d[180, 200] == 254 and 0 < d[110, 120] < 254      # labels held, interior interpolated


@pytest.mark.parametrize("how,unit", [("sor", "sweeps"), ("mg", "cycles")])
def test_harness_refine_extension(tmp_path, how, unit):
    """--refine: rtdd_refine_depth after the estimate, reported on stdout; the depth map stays a valid one."""
    import re
    g = load(NAMES[2])
    _write_pnm(tmp_path / "img.ppm", g["bgr"][..., ::-1])
    _write_pnm(tmp_path / "ann.pgm", g["annotation"])
    out = subprocess.check_output([BIN, "-i", str(tmp_path / "img.ppm"), "-a", str(tmp_path / "ann.pgm"), "-o", str(tmp_path) + "/", "--refine", how], text=True)
    m = re.search(r"refine %s: (\d+) %s, residual ([0-9.e+-]+)" % (how, unit), out)
    assert m, out
    if how == "sor":
        assert float(m.group(2)) <= 1e-4
    d = _read_pnm(tmp_path / "DepthMap.pgm")
    lab = g["mask0"] == 255
    assert np.array_equal(d[lab], g["depth_u8"][lab])                  # labels are Dirichlet values: untouched


def test_plain_c_program_drives_an_estimate(tmp_path):
    """tests/c_abi_smoke.c: C99, links only librtdd.so, runs a whole estimate through the C ABI."""
    lib_dir = os.path.join(ROOT, "realtimedepthdiffusion_amd")
    exe = str(tmp_path / "c_abi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "c_abi_smoke.c"), "-L" + lib_dir, "-lrtdd", "-Wl,-rpath," + lib_dir])
    out = subprocess.check_output([exe], text=True)
    assert "c_abi_smoke ok" in out, out
