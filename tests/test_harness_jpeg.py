"""The harness's JPEG reader (harness/jpeg_reader.hpp) -- the reference loads its images with cv::imread (src/main.cpp:93,158) and all
twelve bundled images are JPEGs, nine of them progressive.  cv::imread decodes with libjpeg(-turbo) at its defaults (slow integer
IDCT, fancy upsampling, fixed-point YCbCr -> RGB): all integer algorithms, restated in the reader and checked here BIT FOR BIT
  * on the twelve dataset files as they are (tests/golden/dataset/<name>.jpg) against their decoded copies (<name>.png, written by
    Pillow's libjpeg-turbo when the fixtures were made) -- needs no Pillow at test time,
  * on streams Pillow encodes here: every chroma layout it can write, sequential / progressive / optimised tables, restart
    markers, gray, qualities 1..100, sizes that are not a whole number of blocks or MCUs and components one or two samples wide,
  * and, built with -fsanitize=address,undefined, on damaged files: refused or decoded, never a sanitizer report.
Runs without a GPU: `rtdd_harness --convert` never touches the device."""
import io
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "harness", "rtdd_harness")
DATASET = os.path.join(ROOT, "tests", "golden", "dataset")
NAMES = sorted(f[:-4] for f in os.listdir(DATASET) if f.endswith(".jpg"))


def _read_pnm(path):
    with open(path, "rb") as f:
        magic = f.readline().strip(); w, h = map(int, f.readline().split()); f.readline()
        a = np.frombuffer(f.read(), np.uint8)
    return a.reshape(h, w, 3) if magic == b"P6" else a.reshape(h, w)


@pytest.fixture(scope="module")
def harness():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "harness")])
    return BIN


def _decode(harness, tmp_path, data, gray=False, exe_env=None):
    src = tmp_path / "in.jpg"
    src.write_bytes(data)
    out = tmp_path / ("out.pgm" if gray else "out.ppm")
    subprocess.check_call([harness, "--convert", str(src), str(out)])
    return _read_pnm(out)


def test_the_fixture_has_all_twelve():
    assert len(NAMES) == 12


@pytest.mark.parametrize("name", NAMES)
def test_dataset_jpeg_decodes_to_the_fixture_pixels(tmp_path, harness, name):
    from PIL import Image                                                # (only to read the lossless PNG copy)
    want = np.array(Image.open(os.path.join(DATASET, name + ".png")).convert("RGB"))
    out = tmp_path / "out.ppm"
    subprocess.check_call([harness, "--convert", os.path.join(DATASET, name + ".jpg"), str(out)])
    assert np.array_equal(_read_pnm(out), want)


def _picture(h, w, seed):
    """smooth gradients + an edge + noise: every coefficient band and both signs of chroma get exercised"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.stack([(xx * 255) // max(w - 1, 1), (yy * 255) // max(h - 1, 1), ((xx + yy) * 7) % 256], -1).astype(np.int32)
    a[(xx > w // 3) & (yy > h // 2)] = (250, 10, 30)
    a += rng.integers(-40, 40, a.shape)
    return np.clip(a, 0, 255).astype(np.uint8)


def _encode(img, mode="RGB", **kw):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(img if mode == "RGB" else img[..., 0], mode).save(buf, "JPEG", **kw)
    return buf.getvalue()


def _pillow(data):
    from PIL import Image
    im = Image.open(io.BytesIO(data))
    return np.array(im), im.mode


CASES = []
for sub in ("4:4:4", "4:2:2", "4:2:0"):
    for prog in (False, True):
        CASES.append(dict(size=(67, 93), subsampling=sub, progressive=prog, quality=85))
CASES += [
    dict(size=(64, 96), subsampling="4:2:0", quality=100),                           # a whole number of MCUs, every quantiser 1
    dict(size=(50, 70), subsampling="4:2:0", quality=1),                             # quantisers up to 255
    dict(size=(50, 70), subsampling="4:2:0", quality=30, optimize=True),
    dict(size=(50, 70), subsampling="4:2:2", quality=60, optimize=True, progressive=True),
    dict(size=(131, 77), subsampling="4:2:0", quality=75, restart_marker_blocks=3),
    dict(size=(131, 77), subsampling="4:2:0", quality=75, restart_marker_rows=1, progressive=True),
    dict(size=(131, 77), subsampling="4:4:4", quality=90, restart_marker_blocks=1),
    dict(size=(1, 1), subsampling="4:2:0", quality=90),
    dict(size=(2, 3), subsampling="4:2:0", quality=90),                              # chroma one / two samples wide: no filter, replication
    dict(size=(3, 4), subsampling="4:2:2", quality=90),
    dict(size=(5, 5), subsampling="4:2:0", quality=90),                              # chroma three wide: the filter's first general case
    dict(size=(9, 17), subsampling="4:2:0", quality=90, progressive=True),
    dict(size=(17, 6), subsampling="4:2:2", quality=50),
    dict(size=(16, 16), subsampling="4:2:0", quality=95),
    dict(size=(203, 311), subsampling="4:2:0", quality=92, progressive=True, optimize=True),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(f"{k}={v}" for k, v in c.items()).replace(" ", ""))
def test_pillow_encoded_colour_streams(tmp_path, harness, case):
    kw = dict(case)
    h, w = kw.pop("size")
    data = _encode(_picture(h, w, h * 1000 + w), **kw)
    want, mode = _pillow(data)
    assert mode == "RGB"
    assert np.array_equal(_decode(harness, tmp_path, data), want)


@pytest.mark.parametrize("prog", [False, True])
@pytest.mark.parametrize("size", [(40, 61), (8, 8), (1, 9)])
def test_gray_streams(tmp_path, harness, size, prog):
    data = _encode(_picture(size[0], size[1], 5), mode="L", quality=80, progressive=prog)
    want, mode = _pillow(data)
    assert mode == "L"
    assert np.array_equal(_decode(harness, tmp_path, data, gray=True), want)


def test_every_quality_setting(tmp_path, harness):
    """quality 1 .. 100 in steps: every scaling of the standard tables, 4:2:0 progressive and sequential alternating"""
    img = _picture(45, 52, 9)
    for q in list(range(1, 100, 7)) + [100]:
        data = _encode(img, quality=q, subsampling="4:2:0", progressive=bool(q & 1))
        assert np.array_equal(_decode(harness, tmp_path, data), _pillow(data)[0]), q


def test_rgb_jpeg_without_a_colour_transform(tmp_path, harness):
    """three components that ARE red, green and blue (Adobe marker, transform 0 -- or component ids 'R' 'G' 'B'): no YCbCr conversion"""
    from PIL import Image
    img = _picture(33, 47, 2)
    buf = io.BytesIO()
    try:
        Image.fromarray(img, "RGB").save(buf, "JPEG", quality=90, keep_rgb=True)
    except (TypeError, OSError, ValueError):
        pytest.skip("this Pillow cannot write RGB JPEGs")
    want, _ = _pillow(buf.getvalue())
    assert np.array_equal(_decode(harness, tmp_path, buf.getvalue()), want)


def test_what_is_not_read_is_refused(tmp_path, harness):
    """CMYK, and anything that is not a JPEG, are refused with exit code 2 -- never guessed at"""
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(np.zeros((8, 8, 4), np.uint8), "CMYK").save(buf, "JPEG")
    for name, data in (("cmyk", buf.getvalue()), ("png_named_jpg", b"\x89PNG\r\n\x1a\n" + b"\0" * 64), ("empty", b"")):
        (tmp_path / "in.jpg").write_bytes(data)
        r = subprocess.run([harness, "--convert", str(tmp_path / "in.jpg"), str(tmp_path / "out.ppm")], capture_output=True, text=True)
        assert r.returncode == 2, (name, r.returncode, r.stderr)


def test_jpeg_code_under_address_and_ub_sanitizers(tmp_path):
    """Damaged JPEGs -- truncated anywhere, bit-flipped in headers, tables and entropy data, lying about their size, tables missing --
    through a build with -fsanitize=address,undefined: decoded (a flip in the entropy data only changes pixels) or refused, never a
    sanitizer report and never a hang."""
    exe = str(tmp_path / "rtdd_harness_asan")
    lib_dir = os.path.join(ROOT, "realtimedepthdiffusion_amd")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-I" + os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "harness", "rtdd_harness.cpp"), "-L" + lib_dir, "-lrtdd", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-lz"]
    if subprocess.run(cmd, capture_output=True).returncode != 0:
        pytest.skip("no sanitizer runtime for g++ here")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    rng = np.random.default_rng(4)
    img = _picture(61, 83, 1)
    streams = {"seq420": _encode(img, quality=80, subsampling="4:2:0"), "prog420": _encode(img, quality=80, subsampling="4:2:0", progressive=True),
               "rst": _encode(img, quality=80, subsampling="4:2:2", restart_marker_blocks=2), "gray": _encode(img, mode="L", quality=70, progressive=True)}
    bad = {}
    for key, good in streams.items():
        (tmp_path / "good.jpg").write_bytes(good)
        out = tmp_path / ("good.pgm" if key == "gray" else "good.ppm")
        r = subprocess.run([exe, "--convert", str(tmp_path / "good.jpg"), str(out)], env=env, capture_output=True, text=True)
        assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (key, r.stderr[-2000:])
        assert np.array_equal(_read_pnm(out), _pillow(good)[0]), key
        for cut in (3, 20, len(good) // 3, len(good) // 2, len(good) - 2):
            bad[f"{key}_cut{cut}"] = good[:cut]
        sos = good.index(b"\xff\xda")
        for k in range(10):                                              # flips in the headers and tables ...
            b = bytearray(good); pos = int(rng.integers(2, sos + 12)); b[pos] ^= 1 << int(rng.integers(0, 8)); bad[f"{key}_hdr{k}"] = bytes(b)
        for k in range(10):                                              # ... and in the entropy-coded data
            b = bytearray(good); pos = int(rng.integers(sos + 12, len(b))); b[pos] ^= 1 << int(rng.integers(0, 8)); bad[f"{key}_ecs{k}"] = bytes(b)
        sof = good.index(b"\xff\xc2") if b"\xff\xc2" in good[:sos] else good.index(b"\xff\xc0")
        huge = bytearray(good); huge[sof + 5:sof + 9] = b"\xff\xff\xff\xff"; bad[f"{key}_lying_size"] = bytes(huge)
        dht = good.index(b"\xff\xc4")
        nodht = bytearray(good); nodht[dht + 1] = 0xFE; bad[f"{key}_no_first_dht"] = bytes(nodht)      # the table segment becomes a comment
        dqt = good.index(b"\xff\xdb")
        nodqt = bytearray(good); nodqt[dqt + 1] = 0xFE; bad[f"{key}_no_dqt"] = bytes(nodqt)
    for name, data in bad.items():
        (tmp_path / "bad.jpg").write_bytes(data)
        r = subprocess.run([exe, "--convert", str(tmp_path / "bad.jpg"), str(tmp_path / "out.ppm")], env=env, capture_output=True, text=True, timeout=120)
        assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (name, r.stderr[-2000:])
        assert r.returncode in (0, 2, 5), (name, r.returncode, r.stderr[-500:])
