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


def test_png_code_under_address_and_ub_sanitizers(tmp_path):
    """The harness's file readers parse untrusted bytes (PNG chunks, zlib streams, PNM headers).  A build of the host program with
    -fsanitize=address,undefined (CPU only: --convert never touches the device) must convert good files cleanly and REJECT damaged
    ones -- truncated, bit-flipped, lying about their size -- with an error, never with a sanitizer report."""
    from PIL import Image
    exe = str(tmp_path / "rtdd_harness_asan")
    lib_dir = os.path.join(ROOT, "realtimedepthdiffusion_amd")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-I" + os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "harness", "rtdd_harness.cpp"), "-L" + lib_dir, "-lrtdd", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-lz"]
    if subprocess.run(cmd, capture_output=True).returncode != 0:
        pytest.skip("no sanitizer runtime for g++ here")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    Image.fromarray(img, "RGB").save(tmp_path / "good.png")
    r = subprocess.run([exe, "--convert", str(tmp_path / "good.png"), str(tmp_path / "good.ppm")], env=env, capture_output=True, text=True)
    assert r.returncode == 0 and "Sanitizer" not in r.stderr, r.stderr
    assert np.array_equal(_read_pnm(tmp_path / "good.ppm"), img)
    good = (tmp_path / "good.png").read_bytes()
    bad = {"truncated": good[: len(good) // 2], "no_iend": good[:-12], "empty": b"", "signature_only": good[:8]}
    for k in range(12):                                                  # bit flips all over the file (CRCs are not checked: the data must fail safely)
        b = bytearray(good); pos = int(rng.integers(8, len(b))); b[pos] ^= 1 << int(rng.integers(0, 8)); bad[f"flip{k}"] = bytes(b)
    huge = bytearray(good); huge[16:24] = (70000).to_bytes(4, "big") + (70000).to_bytes(4, "big"); bad["lying_size"] = bytes(huge)
    bad["pnm_short"] = b"P6\n4000 4000\n255\n" + b"\x00" * 100
    for name, data in bad.items():
        path = tmp_path / (name + (".ppm" if name.startswith("pnm") else ".png"))
        path.write_bytes(data)
        r = subprocess.run([exe, "--convert", str(path), str(tmp_path / "out.ppm")], env=env, capture_output=True, text=True, timeout=60)
        assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (name, r.stderr[-2000:])
        assert r.returncode in (0, 2, 5), (name, r.returncode, r.stderr[-500:])   # converted (a flip in ancillary bytes) or refused -- never crashed
