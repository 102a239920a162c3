"""The C++ headless harness (harness/rtdd_harness, host code over the C ABI) end to end on a golden crop (-m gpu)."""
import json
import os
import subprocess
import sys

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


@pytest.mark.parametrize("ext", ["jpg", "pgm"])
def test_harness_one_channel_image_becomes_three(tmp_path, ext):
    """A gray JPEG (decoded by the harness's own reader) or a PGM as the IMAGE: cv::imread's default flag hands main.cpp three equal channels
    (src/main.cpp:93), so the estimate is the oracle cascade's on B = G = R = the file's pixels."""
    from PIL import Image
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "harness")])
    g = load(NAMES[0])
    gray = np.ascontiguousarray(g["bgr"][..., 1])
    if ext == "jpg":
        Image.fromarray(gray, "L").save(tmp_path / "img.jpg", quality=90, progressive=True)
        gray = np.array(Image.open(tmp_path / "img.jpg"))              # (what libjpeg makes of it: tests/test_harness_jpeg.py checks the reader against that)
        assert gray.ndim == 2
    else:
        _write_pnm(tmp_path / "img.pgm", gray)
    _write_pnm(tmp_path / "ann.pgm", g["annotation"])
    subprocess.check_output([BIN, "-i", str(tmp_path / ("img." + ext)), "-a", str(tmp_path / "ann.pgm"), "-o", str(tmp_path) + "/"], text=True)
    ref = _oracle_cascade({"bgr": np.repeat(gray[..., None], 3, 2), "annotation": g["annotation"]})
    assert np.array_equal(_read_pnm(tmp_path / "DepthMap.pgm"), ref.depth_u8)


def _oracle_cascade(g, paints=(), annotation=True, estimates=1, iters=1000):
    """The harness's call sequence restated over the oracle (tests/cascade_ref.py): decode, brush samples, `estimates` estimates."""
    import oracle
    from cascade_ref import Cascade
    lut = oracle.load_weights(0.4)
    c = Cascade(oracle, g["bgr"], g["annotation"] if annotation else None, lut, 1, threads=4)
    for x, y, label, radius in paints:                                 # GPUPaintImage, src/GPUImageProcessing.cu:51-70 (main.cpp:55-57)
        oracle.paint_image(x, y, label, radius, c.edited[0], c.scribble[0])
    for _ in range(estimates):
        c.estimate(iters)
    return c


def test_harness_live_mode_is_the_warm_started_estimate(tmp_path):
    """--live N (src/main.cpp:232 `live`): N estimates of one image, each warm-started from the one before (the depth
    pyramid and the coarse annotation levels persist); the written map is the N-th estimate of the restated cascade,
    and differs from the single cold estimate."""
    g = load(NAMES[0])
    _write_pnm(tmp_path / "img.ppm", g["bgr"][..., ::-1])
    _write_pnm(tmp_path / "ann.pgm", g["annotation"])
    out = subprocess.check_output([BIN, "-i", str(tmp_path / "img.ppm"), "-a", str(tmp_path / "ann.pgm"), "-o", str(tmp_path) + "/", "--live", "3", "--iters", "300"], text=True)
    assert "3 estimate(s) on 1 device(s)" in out
    d = _read_pnm(tmp_path / "DepthMap.pgm")
    third = _oracle_cascade(g, estimates=3, iters=300)
    assert np.array_equal(d, third.depth_u8)
    first = _oracle_cascade(g, estimates=1, iters=300)
    assert not np.array_equal(first.depth[0], third.depth[0])          # the warm start really changes something at 300 sweeps


def test_harness_live_frames_are_pipelined_and_each_is_the_oracles(tmp_path):
    """--live 4 --write-all: the four frames go through rtdd_live_submit / rtdd_live_wait, two in flight (annotation upload, estimate,
    map download per frame: the region src/main.cpp:234-293 clocks); EVERY frame's map is the oracle's n-th warm-started estimate."""
    g = load(NAMES[1])
    _write_pnm(tmp_path / "img.ppm", g["bgr"][..., ::-1])
    _write_pnm(tmp_path / "ann.pgm", g["annotation"])
    paint = (60, 70, 192, 9)
    out = subprocess.check_output([BIN, "-i", str(tmp_path / "img.ppm"), "-a", str(tmp_path / "ann.pgm"), "-o", str(tmp_path) + "/", "--live", "4", "--iters", "250",
                                   "--write-all", "--paint", "%d,%d,%d,%d" % paint], text=True)
    assert "Live:" in out and "frames/s" in out and "two frames in flight" in out, out
    import oracle
    from cascade_ref import Cascade
    c = Cascade(oracle, g["bgr"], g["annotation"], oracle.load_weights(0.4), 1, threads=4)
    oracle.paint_image(*paint, c.edited[0], c.scribble[0])
    for n in range(4):
        c.estimate(250)
        assert np.array_equal(_read_pnm(tmp_path / f"DepthMap_{n}.pgm"), c.depth_u8), f"frame {n}"
    assert np.array_equal(_read_pnm(tmp_path / "DepthMap.pgm"), c.depth_u8)


@pytest.mark.parametrize("effect", ["defocus", "haze"])
def test_harness_live_frames_with_a_sticky_effect(tmp_path, effect):
    """--live 3 --effect X: the reference's frame with a sticky effect (src/main.cpp:190-230 beside :232-295) through
    rtdd_live_submit_ex, pipelined: the written artistic image is the oracle's effect on the oracle's third warm-started estimate."""
    g = load(NAMES[0])
    _write_pnm(tmp_path / "img.ppm", g["bgr"][..., ::-1])
    _write_pnm(tmp_path / "ann.pgm", g["annotation"])
    out = subprocess.check_output([BIN, "-i", str(tmp_path / "img.ppm"), "-a", str(tmp_path / "ann.pgm"), "-o", str(tmp_path) + "/", "--live", "3", "--iters", "300",
                                   "--effect", effect], text=True)
    assert "Live:" in out and "two frames in flight" in out, out
    import oracle
    third = _oracle_cascade(g, estimates=3, iters=300)
    assert np.array_equal(_read_pnm(tmp_path / "DepthMap.pgm"), third.depth_u8)
    want = oracle.defocus(g["bgr"], third.depth[0], threads=4) if effect == "defocus" else oracle.haze(g["bgr"], third.depth[0], 1)
    assert np.array_equal(_read_pnm(tmp_path / "ArtisticEffect.ppm"), want[..., ::-1])


def test_harness_painting_into_a_live_view(tmp_path):
    """--live 5 --paint-at 2:... --paint-at 4:...: strokes in front of frames 2 and 4 (main.cpp:46-62: the mouse callback paints the
    device images and downloads them into the host's, the next frame uploads them, :236-237).  Every frame == the oracle cascade with
    the same strokes at the same places in the sequence of warm-started estimates."""
    g = load(NAMES[0])
    _write_pnm(tmp_path / "img.ppm", g["bgr"][..., ::-1])
    _write_pnm(tmp_path / "ann.pgm", g["annotation"])
    strokes = {2: (50, 60, 254, 11), 4: (180, 200, 0, 7)}
    args = [BIN, "-i", str(tmp_path / "img.ppm"), "-a", str(tmp_path / "ann.pgm"), "-o", str(tmp_path) + "/", "--live", "5", "--iters", "200", "--write-all"]
    for f, st in strokes.items():
        args += ["--paint-at", "%d:%d,%d,%d,%d" % ((f,) + st)]
    subprocess.check_output(args, text=True)
    import oracle
    from cascade_ref import Cascade
    c = Cascade(oracle, g["bgr"], g["annotation"], oracle.load_weights(0.4), 1, threads=4)
    for n in range(5):
        if n in strokes:
            oracle.paint_image(*strokes[n], c.edited[0], c.scribble[0])
        c.estimate(200)
        assert np.array_equal(_read_pnm(tmp_path / f"DepthMap_{n}.pgm"), c.depth_u8), f"frame {n}"
    assert np.array_equal(_read_pnm(tmp_path / "AnnotatedImage.ppm"), c.edited[0][..., ::-1])


def test_harness_batch_and_paint(tmp_path):
    """--paint (the mouse-drag brush, main.cpp:46-62) against GPUPaintImage's restatement + the restated cascade, every pixel;
    --batch B treats every image as independent: B = 3 on one device writes the same map as B = 1 (no warm start leaks in)."""
    g = load(NAMES[0])
    _write_pnm(tmp_path / "img.ppm", g["bgr"][..., ::-1])
    paints = [(40, 40, 0, 9), (200, 180, 254, 9), (128, 10, 128, 5)]
    args = [BIN, "-i", str(tmp_path / "img.ppm"), "-o", str(tmp_path) + "/", "--iters", "200"]
    for p in paints:
        args += ["--paint", "%d,%d,%d,%d" % p]
    out = subprocess.check_output(args + ["--batch", "3", "--devices", "1"], text=True)
    assert "3 estimate(s) on 1 device(s)" in out
    d3 = _read_pnm(tmp_path / "DepthMap.pgm")
    want = _oracle_cascade(g, paints=paints, annotation=False, estimates=1, iters=200)
    assert np.array_equal(d3, want.depth_u8)
    assert np.array_equal(_read_pnm(tmp_path / "AnnotatedImage.ppm"), want.edited[0][..., ::-1])       # main.cpp:298-303: editedImage[0] (BGR in memory)
    assert d3[40, 40] == 0 and d3[180, 200] == 254
    subprocess.check_output(args + ["--batch", "1"], text=True)
    assert np.array_equal(_read_pnm(tmp_path / "DepthMap.pgm"), d3)


def test_harness_batch_of_eight_writes_eight_identical_maps(tmp_path):
    """SURVEY 4, multi-GPU tier at N = 1: --batch 8 --write-all writes the depth map of EVERY one of the eight independent
    estimates; each equals the single-image result bitwise (and the golden map)."""
    g = load(NAMES[1])
    _write_pnm(tmp_path / "img.ppm", g["bgr"][..., ::-1])
    _write_pnm(tmp_path / "ann.pgm", g["annotation"])
    args = [BIN, "-i", str(tmp_path / "img.ppm"), "-a", str(tmp_path / "ann.pgm"), "-o", str(tmp_path) + "/"]
    subprocess.check_output(args, text=True)
    single = _read_pnm(tmp_path / "DepthMap.pgm")
    assert np.array_equal(single, g["depth_u8"])
    out = subprocess.check_output(args + ["--batch", "8", "--devices", "8", "--write-all"], text=True)   # --devices is capped at what the box has
    assert "8 estimate(s) on" in out
    for b in range(8):
        assert np.array_equal(_read_pnm(tmp_path / f"DepthMap_{b}.pgm"), single), f"image {b} of the batch"


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


def test_no_state_leaks_between_contexts_of_one_device(tmp_path):
    """Several contexts on ONE device, first inside one process, interleaved -- a context that has healed a time-out (persistence off,
    status word set and cleared), one whose defocus went to the table after a depth that is no depth, one with forced options -- must
    leave a bystander context's flag epoch, options, counters and BITS alone; then, as a process tree, the harness with --devices 1
    --batch 8 and bench.py --gpus 1 on a fixed batch, back to back, each verified (one context + stream per GPU, nothing
    shared but the device)."""
    import oracle
    import realtimedepthdiffusion_amd as rt
    from gpu_util import assert_bit_equal, down, up
    from realtimedepthdiffusion_amd.synth import make_problem
    rows, cols = 1080, 1920
    p = make_problem(rows, cols, seed=21)
    lut = oracle.load_weights(0.4)
    want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 48, 0, 0, lut, 1, threads=oracle.max_threads())
    m, g = up(p["mask"]), up(p["gray"])
    rgb = np.random.default_rng(3).integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    blur = oracle.defocus(rgb, want, threads=oracle.max_threads())

    def check(c, what):
        d = up(p["depth"])
        c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 48, 0.0, 0)
        art = up(np.zeros_like(rgb))
        c.GPUSimulateDefocus(up(rgb), d, art, rows, cols)
        c.synchronize()
        i = c.last_solve_info()
        assert (i.persistent, c.get_option(rt.OPT_PERSISTENT), c.get_option(rt.OPT_TIMEOUT_HEALS), c.get_option(rt.OPT_DEFOCUS_LAST_PATH)) == (1, 1, 0, 2), (what, i.describe())
        assert_bit_equal(down(d), want, what)
        assert np.array_equal(down(art), blur), what

    with rt.Context(0) as a, rt.Context(0) as b, rt.Context(0) as c3:
        for c in (a, b, c3):
            c.GPUAllocateDeviceMemory(rows, cols, 1); c.GPULoadWeights(0.4)
        check(a, "bystander, before")
        # b: a timed-out persistent launch, healed
        b.set_option(rt.OPT_DEBUG_POLL_LIMIT_US, 3000); b.set_option(rt.OPT_DEBUG_WITHHOLD_TILE, 5)
        d = up(p["depth"]); b.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 48, 0.0, 0)
        d_a = up(p["depth"]); a.GPUMatrixFreeSolver(d_a, m, g, rows, cols, 0.4, 48, 0.0, 0)          # queued while b's launch is failing
        b.synchronize(); a.synchronize()
        assert b.get_option(rt.OPT_TIMEOUT_HEALS) == 1 and b.get_option(rt.OPT_PERSISTENT) == 0
        assert_bit_equal(down(d), want, "the healed context"); assert_bit_equal(down(d_a), want, "the bystander, during")
        # c3: defocus sent to the table by a depth that is no depth; a forced status 2 reported
        crazy = up(np.full((rows, cols), 3.0e5, np.float32)); art = up(np.zeros_like(rgb))
        c3.GPUSimulateDefocus(up(rgb), crazy, art, rows, cols); c3.synchronize()
        c3.GPUSimulateDefocus(up(rgb), crazy, art, rows, cols); c3.synchronize()
        assert c3.get_option(rt.OPT_DEFOCUS_LAST_PATH) == 1
        c3.set_option(rt.OPT_DEBUG_FORCE_STATUS, 2)
        d3 = up(p["depth"]); c3.GPUMatrixFreeSolver(d3, m, g, rows, cols, 0.4, 48, 0.0, 0)
        with pytest.raises(rt.RtddError):
            c3.synchronize()
        check(a, "bystander, after")
    # the same as a process tree: the harness's batch, then the bench's, on the same device
    gl = load(NAMES[1])
    _write_pnm(tmp_path / "img.ppm", gl["bgr"][..., ::-1]); _write_pnm(tmp_path / "ann.pgm", gl["annotation"])
    out = subprocess.check_output([BIN, "-i", str(tmp_path / "img.ppm"), "-a", str(tmp_path / "ann.pgm"), "-o", str(tmp_path) + "/", "--devices", "1", "--batch", "8", "--write-all"], text=True)
    assert "8 estimate(s) on 1 device(s)" in out
    single = _oracle_cascade(gl).depth_u8
    for n in range(8):
        assert np.array_equal(_read_pnm(tmp_path / f"DepthMap_{n}.pgm"), single), n
    line = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "batch8_270x480x200", "--steps", "2", "--warmup", "1",
                                    "--no-cpu-baseline", "--verify"], text=True, stderr=subprocess.DEVNULL).strip().splitlines()[-1]
    v = json.loads(line)["verified"]
    assert v["images_differing_all_ranks"] == 0 and len(v["rank0_images"]) == 8, v
