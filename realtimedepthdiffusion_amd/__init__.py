"""realtimedepthdiffusion_amd -- host-side Python mirror of librtdd.so's C ABI.

The product is the HIP library (csrc/ -> librtdd.so, declared in include/rtdd.h).  This
module is plumbing: it loads the library with ctypes and mirrors the reference's interface
for the hot path -- the ten ``GPU*`` functions of /root/reference/include/GPUSolver.h:6-10,
GPUImageProcessing.h:4-10 and GPUDepthEffect.h:4-9 -- with the same names, argument order
and meaning, so parity tests read like calls into the reference.  Image arguments are
*pitched device buffers*: anything with ``data_ptr()`` and ``stride()`` (a torch CUDA tensor
whose rows may be padded) or an explicit ``(ptr, pitch_bytes)`` pair.

There is NO CPU fallback: if librtdd.so is missing or no HIP device is present every entry
point raises.  Nothing here imports ``oracle``.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("RTDD_LIBRARY") or os.path.join(_HERE, "librtdd.so")     # RTDD_LIBRARY: developer knob for A/B builds (scripts/build_variant.sh)
_CSRC = os.path.join(_HERE, "csrc")

RTDD_OK = 0
METHOD_CHEBYSHEV_JACOBI = 0
METHOD_RED_BLACK_GS = 1
METHOD_MULTIGRID = 2
METHOD_AUTO = 3
RELAXATION_AUTO = -1.0                     # rtdd_solve_params.relaxation: SOR cycles (include/rtdd.h)
OPT_FP_CONTRACT, OPT_SWEEP_KERNEL, OPT_TEMPORAL_DEPTH, OPT_DEFOCUS_PATH, OPT_ROWS_PER_WAVE, OPT_TILE, OPT_PERSISTENT = 0, 1, 2, 3, 4, 5, 6
OPT_DEBUG_WITHHOLD_TILE, OPT_DEBUG_POLL_LIMIT_US = 7, 8
OPT_AUTO_CYCLE_FIXED_NS, OPT_AUTO_CYCLE_FS_PER_PX, OPT_AUTO_SWEEP_FS_PER_PX, OPT_AUTO_SWEEP_FLOOR_NS = 9, 10, 11, 12
OPT_DEBUG_FORCE_STATUS = 13
OPT_TIMEOUT_HEALS = 14                     # read only
OPT_DEFOCUS_LAST_PATH = 15                 # read only: 1 the global table, 2 the tile kernel
OPT_TIMEOUT_HEAL, OPT_PERSISTENT_REARM_AFTER, OPT_PERSISTENT_SUSPENDED, OPT_PENDING_CALLS = 16, 17, 18, 19
OPT_LIVE_ZERO_COPY = 20
OPT_ANNOTATION_LDS = 21
OPT_DEFOCUS_STRIPS = 22
OPT_DEFOCUS_LAST_SLICES = 23              # read only
OPT_DEFOCUS_SLICE_MB = 24
RTDD_ERR_TIMEOUT = 6

# every symbol include/rtdd.h declares (checked by tests/test_abi.py against the header)
C_ABI_SYMBOLS = [
    "rtdd_ctx_create", "rtdd_ctx_destroy", "rtdd_ctx_set_stream", "rtdd_ctx_synchronize", "rtdd_set_option",
    "rtdd_get_option", "rtdd_last_error", "rtdd_status_string", "rtdd_version", "rtdd_allocate", "rtdd_free",
    "rtdd_load_weights", "rtdd_matrix_free_solver", "rtdd_solve_ex", "rtdd_last_solve_info", "rtdd_multigrid_level", "rtdd_index_to_weight", "rtdd_convert_to_float",
    "rtdd_pyrdown_annotation", "rtdd_paint_image", "rtdd_simulate_defocus", "rtdd_simulate_desaturation",
    "rtdd_simulate_haze", "rtdd_profile_enable", "rtdd_profile_get",
    "rtdd_pyramid_levels", "rtdd_pyramid_create", "rtdd_pyramid_destroy", "rtdd_pyramid_set_image",
    "rtdd_pyramid_set_annotation", "rtdd_pyramid_image", "rtdd_pyramid_annotation_changed", "rtdd_estimate_depth", "rtdd_refine_depth", "rtdd_bgr2gray", "rtdd_pyrdown_gray",
    "rtdd_pyrup_depth", "rtdd_depth_to_u8", "rtdd_upload", "rtdd_download",
    "rtdd_live_submit", "rtdd_live_wait", "rtdd_live_pending", "rtdd_host_alloc", "rtdd_host_free",
    "rtdd_pyramid_create_batch", "rtdd_pyramid_select", "rtdd_pyramid_batch", "rtdd_estimate_depth_batch", "rtdd_pyramid_level_info", "rtdd_live_submit_ex",
]
IMG_ORIGINAL, IMG_GRAY, IMG_SCRIBBLE, IMG_EDITED, IMG_DEPTH, IMG_DEPTH_U8, IMG_ARTISTIC = range(7)
EFFECT_NONE, EFFECT_DEFOCUS, EFFECT_DESATURATION, EFFECT_HAZE = range(4)
# Itanium-mangled names of the reference's ten free functions
DROPIN_SYMBOLS = [
    "_Z23GPUAllocateDeviceMemoryiii", "_Z19GPUFreeDeviceMemoryi", "_Z14GPULoadWeightsf",
    "_Z19GPUMatrixFreeSolverPfmPhmS0_miififi", "_Z17GPUConvertToFloatPhmPfmS_mii",
    "_Z20GPUPyrDownAnnotationPhmS_miiS_mS_mii", "_Z13GPUPaintImageiiiiPhmS_mii",
    "_Z18GPUSimulateDefocusPhmPfmS_mii", "_Z23GPUSimulateDesaturationPhmS_mPfmS_mii", "_Z15GPUSimulateHazePhmPfmS_mii",
]


class RtddError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"rtdd status {status}: {message}")
        self.status = status


class SolveParams(C.Structure):
    _fields_ = [("method", C.c_int), ("maxIterations", C.c_int), ("tolerance", C.c_float), ("checkEvery", C.c_int), ("relaxation", C.c_float)]


class SolveInfo(C.Structure):
    _fields_ = [("iterations", C.c_int), ("residual", C.c_float), ("cycles", C.c_int), ("kernel", C.c_int), ("tile", C.c_int),
                ("temporal_depth", C.c_int), ("persistent", C.c_int), ("fp_contract", C.c_int), ("launches", C.c_int)]

    def describe(self):
        return (f"kernel {self.kernel} tile {self.tile} depth {self.temporal_depth} persistent {self.persistent} contract {self.fp_contract} "
                f"launches {self.launches} iterations {self.iterations} cycles {self.cycles} residual {self.residual:.3g}")


class Profile(C.Structure):
    _fields_ = [("sweep_ms", C.c_double), ("launches", C.c_int), ("sweeps", C.c_int),
                ("prepare_ms", C.c_double), ("finish_ms", C.c_double)]


def build(force=False):
    """Compile librtdd.so for gfx950 (hipcc cross-compiles without a GPU).

    The persistent kernels' hand-off reads its halo with sc1 loads and NO agent-scope acquire; that is only right if the compiler put
    nothing between those loads and the wait behind them (csrc/sweep_common.hpp).  scripts/isa_check.py checks exactly that on the
    objects just built -- every path, spills included.  If the check fails, or cannot run (no llvm-objdump), the two kernels are
    rebuilt with -DRTDD_EXCHANGE_ACQUIRE=1 (one acquire per exchange, plain loads: ~0.8 us per exchange slower, no such dependence) and
    a warning says so: a compiler bump can cost speed, never silently a wrong depth map."""
    import sys
    args = ["make", "-C", _CSRC, "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    stamp = os.path.join(_CSRC, ".exchange_variant")
    try:
        sys.path.insert(0, os.path.join(os.path.dirname(_HERE), "scripts"))
        import isa_check
        isa_check.check_build(_CSRC)
        variant = "sc1"
    except Exception as e:      # AssertionError (the check failed), FileNotFoundError (no objdump), ...
        already = os.path.exists(stamp) and open(stamp).read().strip() == "acquire" and os.path.getmtime(stamp) >= os.path.getmtime(os.path.join(_CSRC, "sweep_blocked.o"))
        if not already:
            print(f"[rtdd build] the structural check of the no-acquire hand-off did not pass ({type(e).__name__}: {e}): rebuilding the persistent "
                  "kernels with -DRTDD_EXCHANGE_ACQUIRE=1", file=sys.stderr)
            for f in ("sweep_blocked.o", "rbgs_blocked.o"):
                if os.path.exists(os.path.join(_CSRC, f)):
                    os.remove(os.path.join(_CSRC, f))
            subprocess.check_call(["make", "-C", _CSRC, "-j4", "CXXFLAGS_EXTRA=-DRTDD_EXCHANGE_ACQUIRE=1"], stdout=subprocess.DEVNULL)
        variant = "acquire"
    with open(stamp, "w") as f:
        f.write(variant + "\n")
    return _SO


_lib = None


def lib():
    """The loaded librtdd.so.  Raises (never falls back) when the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise RtddError(-1, f"{_SO} is not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(there is no CPU fallback)")
        try:
            # torch wheels bundle their own libamdhip64 (SONAME libamdhip64.so.7, NEEDED as plain
            # "libamdhip64.so").  If librtdd.so pulled in /opt/rocm's copy first, torch would load a
            # SECOND HIP runtime into the process and find no GPUs.  Loading torch first makes the
            # dynamic linker satisfy librtdd's libamdhip64.so.7 with the runtime torch already holds.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(_SO)
        L.rtdd_last_error.restype = C.c_char_p
        L.rtdd_status_string.restype = C.c_char_p
        for name in C_ABI_SYMBOLS:
            getattr(L, name)        # fail at load time, not at first use, if a symbol is missing
        if L.rtdd_version() // 100 != 2:    # SolveInfo below is the 36-byte rtdd_solve_info of ABI 2xx
            raise RtddError(-1, f"{_SO} has ABI version {L.rtdd_version()}, this binding needs 2xx: rebuild")
        _lib = L
    return _lib


def _img(t):
    """(pointer, pitch in bytes) of a pitched device image."""
    if isinstance(t, tuple):
        return C.c_void_p(t[0]), C.c_size_t(t[1])
    assert t.stride(-1) == 1 or (t.dim() == 3 and t.stride(2) == 1 and t.stride(1) == t.shape[2]), "pixels must be contiguous within a row"
    return C.c_void_p(t.data_ptr()), C.c_size_t(t.stride(0) * t.element_size())


class Context:
    """One solver context per GPU (handle of the C ABI).  Methods carry the reference's names."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        rc = lib().rtdd_ctx_create(C.c_int(device), C.byref(self._h))
        if rc != RTDD_OK:
            raise RtddError(rc, lib().rtdd_status_string(rc).decode())
        self.device = device
        self.last_cycles = 0                    # V-cycles of the last solve_ex / refine_depth (rtdd_solve_info.cycles)
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().rtdd_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != RTDD_OK:
            raise RtddError(rc, lib().rtdd_status_string(rc).decode() + " -- " + lib().rtdd_last_error(self._h).decode())

    # ---- context plumbing
    def set_stream(self, stream):
        """stream: a raw hipStream_t value (e.g. torch.cuda.current_stream().cuda_stream) or 0."""
        self._check(lib().rtdd_ctx_set_stream(self._h, C.c_void_p(int(stream))))

    def synchronize(self):
        self._check(lib().rtdd_ctx_synchronize(self._h))

    def set_option(self, key, value):
        self._check(lib().rtdd_set_option(self._h, C.c_int(key), C.c_int(value)))

    def get_option(self, key):
        v = C.c_int()
        self._check(lib().rtdd_get_option(self._h, C.c_int(key), C.byref(v)))
        return v.value

    def auto_model(self, rows, cols):
        """(seconds RTDD_METHOD_AUTO prices its SOR-cycle alternative at, seconds per V-cycle): csrc/api.cpp Solve::automatic(),
        from the RTDD_OPT_AUTO_* constants the library holds."""
        px = float(rows * cols)
        sweep = max(px * self.get_option(OPT_AUTO_SWEEP_FS_PER_PX) * 1e-15, self.get_option(OPT_AUTO_SWEEP_FLOOR_NS) * 1e-9)
        sor = (((max(rows, cols) + 1) // 2) * 1.25 + 20.0) * sweep
        cycle = self.get_option(OPT_AUTO_CYCLE_FIXED_NS) * 1e-9 + px * self.get_option(OPT_AUTO_CYCLE_FS_PER_PX) * 1e-15
        return sor, cycle

    def profile_enable(self, on=True):
        self._check(lib().rtdd_profile_enable(self._h, C.c_int(1 if on else 0)))

    def profile(self):
        p = Profile()
        self._check(lib().rtdd_profile_get(self._h, C.byref(p)))
        return p

    # ---- include/GPUSolver.h
    def GPUAllocateDeviceMemory(self, rows, cols, levels):
        self._check(lib().rtdd_allocate(self._h, C.c_int(rows), C.c_int(cols), C.c_int(levels)))

    def GPUFreeDeviceMemory(self, levels=0):
        self._check(lib().rtdd_free(self._h))

    def GPULoadWeights(self, beta):
        self._check(lib().rtdd_load_weights(self._h, C.c_float(beta)))

    def GPUMatrixFreeSolver(self, depthImage, scribbleImage, grayImage, rows, cols, beta, maxIterations, tolerance, level):
        dp, dpitch = _img(depthImage); sp, spitch = _img(scribbleImage); gp, gpitch = _img(grayImage)
        self._check(lib().rtdd_matrix_free_solver(self._h, dp, dpitch, sp, spitch, gp, gpitch, C.c_int(rows), C.c_int(cols),
                                                  C.c_float(beta), C.c_int(maxIterations), C.c_float(tolerance), C.c_int(level)))

    def solve_ex(self, depthImage, scribbleImage, grayImage, rows, cols, level, method=METHOD_CHEBYSHEV_JACOBI,
                 maxIterations=1000, tolerance=0.0, checkEvery=0, relaxation=0.0):
        dp, dpitch = _img(depthImage); sp, spitch = _img(scribbleImage); gp, gpitch = _img(grayImage)
        params = SolveParams(method, maxIterations, tolerance, checkEvery, relaxation)
        info = SolveInfo()
        self._check(lib().rtdd_solve_ex(self._h, dp, dpitch, sp, spitch, gp, gpitch, C.c_int(rows), C.c_int(cols), C.c_int(level),
                                        C.byref(params), C.byref(info)))
        self.last_cycles = info.cycles          # V-cycles of the last solve (METHOD_MULTIGRID / METHOD_AUTO)
        return info.iterations, info.residual

    def last_solve_info(self):
        """What the most recent solve on this context actually ran (kernel, tile, depth, persistence, contraction, counts)."""
        info = SolveInfo()
        self._check(lib().rtdd_last_solve_info(self._h, C.byref(info)))
        return info

    def multigrid_level(self, level, which):
        """Diagnostic: plane `which` of hierarchy level `level` after a METHOD_MULTIGRID solve, as a numpy array."""
        import numpy as np
        r = C.c_int(0); c = C.c_int(0)
        self._check(lib().rtdd_multigrid_level(self._h, C.c_int(level), C.c_int(which), None, C.byref(r), C.byref(c)))
        out = np.zeros((r.value, c.value), np.float32)
        self._check(lib().rtdd_multigrid_level(self._h, C.c_int(level), C.c_int(which), out.ctypes.data_as(C.c_void_p), C.byref(r), C.byref(c)))
        return out

    def index_to_weight(self, grayImage, depthImage, index2, level, rows, cols):
        gp, gpitch = _img(grayImage); dp, dpitch = _img(depthImage)
        self._check(lib().rtdd_index_to_weight(self._h, gp, gpitch, dp, dpitch, C.c_void_p(index2.data_ptr()), C.c_int(level),
                                               C.c_int(rows), C.c_int(cols)))

    # ---- include/GPUImageProcessing.h
    def GPUConvertToFloat(self, src, dst, mask, rows, cols):
        s, sp = _img(src); d, dp = _img(dst); m, mp = _img(mask)
        self._check(lib().rtdd_convert_to_float(self._h, s, sp, d, dp, m, mp, C.c_int(rows), C.c_int(cols)))

    def GPUPyrDownAnnotation(self, prevScribbleImage, prevEditedImage, previousRows, previousCols,
                             currScribbleImage, currEditedImage, currentRows, currentCols):
        a, ap = _img(prevScribbleImage); b, bp = _img(prevEditedImage); c, cp = _img(currScribbleImage); d, dp = _img(currEditedImage)
        self._check(lib().rtdd_pyrdown_annotation(self._h, a, ap, b, bp, C.c_int(previousRows), C.c_int(previousCols),
                                                  c, cp, d, dp, C.c_int(currentRows), C.c_int(currentCols)))

    def GPUPaintImage(self, x, y, scribbleColor, scribbleRadius, editedImage, scribbleImage, rows, cols):
        e, ep = _img(editedImage); s, sp = _img(scribbleImage)
        self._check(lib().rtdd_paint_image(self._h, C.c_int(x), C.c_int(y), C.c_int(scribbleColor), C.c_int(scribbleRadius),
                                           e, ep, s, sp, C.c_int(rows), C.c_int(cols)))

    # ---- include/GPUDepthEffect.h
    def GPUSimulateDefocus(self, originalImage, depthImage, artisticImage, rows, cols):
        o, op = _img(originalImage); d, dp = _img(depthImage); a, ap = _img(artisticImage)
        self._check(lib().rtdd_simulate_defocus(self._h, o, op, d, dp, a, ap, C.c_int(rows), C.c_int(cols)))

    def GPUSimulateDesaturation(self, originalImage, grayImage, depthImage, artisticImage, rows, cols):
        o, op = _img(originalImage); g, gp = _img(grayImage); d, dp = _img(depthImage); a, ap = _img(artisticImage)
        self._check(lib().rtdd_simulate_desaturation(self._h, o, op, g, gp, d, dp, a, ap, C.c_int(rows), C.c_int(cols)))

    def GPUSimulateHaze(self, originalImage, depthImage, artisticImage, rows, cols):
        o, op = _img(originalImage); d, dp = _img(depthImage); a, ap = _img(artisticImage)
        self._check(lib().rtdd_simulate_haze(self._h, o, op, d, dp, a, ap, C.c_int(rows), C.c_int(cols)))


    # ---- whole-estimate driver (src/main.cpp:92-155, 232-295)
    def pyramid_create(self, rows, cols):
        self._check(lib().rtdd_pyramid_create(self._h, C.c_int(rows), C.c_int(cols)))
        return int(lib().rtdd_pyramid_levels(C.c_int(rows), C.c_int(cols)))

    def pyramid_create_batch(self, rows, cols, images):
        """`images` pyramids of one size on this context (rtdd_estimate_depth_batch runs them in the same launches)."""
        self._check(lib().rtdd_pyramid_create_batch(self._h, C.c_int(rows), C.c_int(cols), C.c_int(images)))
        return int(lib().rtdd_pyramid_levels(C.c_int(rows), C.c_int(cols)))

    def pyramid_select(self, index):
        self._check(lib().rtdd_pyramid_select(self._h, C.c_int(index)))

    def estimate_depth_batch(self, maxIterations=1000):
        self._check(lib().rtdd_estimate_depth_batch(self._h, C.c_int(maxIterations)))

    def pyramid_level_info(self, level):
        """(SolveInfo, images per sweep launch) of what the most recent estimate ran on pyramid level `level`."""
        info = SolveInfo(); n = C.c_int()
        self._check(lib().rtdd_pyramid_level_info(self._h, C.c_int(level), C.byref(info), C.byref(n)))
        return info, n.value

    def pyramid_destroy(self):
        self._check(lib().rtdd_pyramid_destroy(self._h))

    def pyramid_set_image(self, bgr):
        p, pitch = _img(bgr)
        self._check(lib().rtdd_pyramid_set_image(self._h, p, pitch))

    def pyramid_set_annotation(self, annotation):
        p, pitch = _img(annotation)
        self._check(lib().rtdd_pyramid_set_annotation(self._h, p, pitch))

    def pyramid_annotation_changed(self):
        self._check(lib().rtdd_pyramid_annotation_changed(self._h))

    def pyramid_image(self, kind, level=0):
        """(ptr, pitch_bytes, rows, cols) of a context-owned pyramid image."""
        ptr = C.c_void_p(); pitch = C.c_size_t(); rows = C.c_int(); cols = C.c_int()
        self._check(lib().rtdd_pyramid_image(self._h, C.c_int(kind), C.c_int(level), C.byref(ptr), C.byref(pitch), C.byref(rows), C.byref(cols)))
        return ptr.value, pitch.value, rows.value, cols.value

    def pyramid_download(self, kind, level=0):
        import numpy as np
        ptr, pitch, rows, cols = self.pyramid_image(kind, level)
        dtype, ch = (np.float32, 1) if kind == IMG_DEPTH else (np.uint8, 3 if kind in (IMG_ORIGINAL, IMG_EDITED, IMG_ARTISTIC) else 1)
        out = np.empty((rows, cols, ch) if ch == 3 else (rows, cols), dtype)
        if rows and cols:
            w = cols * ch * out.itemsize
            self._check(lib().rtdd_download(self._h, C.c_void_p(out.ctypes.data), C.c_size_t(w), C.c_void_p(ptr), C.c_size_t(pitch), C.c_size_t(w), C.c_int(rows)))
        return out

    def estimate_depth(self, maxIterations=1000):
        self._check(lib().rtdd_estimate_depth(self._h, C.c_int(maxIterations)))

    # ---- live mode (src/main.cpp:232-295 pipelined): host images are numpy views of page-locked memory (host_image)
    def live_submit(self, scribble, edited, depth_u8, maxIterations=1000):
        sp = (C.c_void_p(scribble.ctypes.data), C.c_size_t(scribble.strides[0])) if scribble is not None else (None, C.c_size_t(0))
        ep = (C.c_void_p(edited.ctypes.data), C.c_size_t(edited.strides[0])) if edited is not None else (None, C.c_size_t(0))
        self._check(lib().rtdd_live_submit(self._h, sp[0], sp[1], ep[0], ep[1], C.c_int(maxIterations),
                                           C.c_void_p(depth_u8.ctypes.data), C.c_size_t(depth_u8.strides[0])))

    def live_submit_ex(self, scribble, edited, depth_u8, effect, artistic, maxIterations=1000):
        """rtdd_live_submit_ex: the frame with a sticky depth effect (EFFECT_*), its artistic image into the host array `artistic`."""
        sp = (C.c_void_p(scribble.ctypes.data), C.c_size_t(scribble.strides[0])) if scribble is not None else (None, C.c_size_t(0))
        ep = (C.c_void_p(edited.ctypes.data), C.c_size_t(edited.strides[0])) if edited is not None else (None, C.c_size_t(0))
        ap = (C.c_void_p(artistic.ctypes.data), C.c_size_t(artistic.strides[0])) if artistic is not None else (None, C.c_size_t(0))
        self._check(lib().rtdd_live_submit_ex(self._h, sp[0], sp[1], ep[0], ep[1], C.c_int(maxIterations),
                                              C.c_void_p(depth_u8.ctypes.data), C.c_size_t(depth_u8.strides[0]), C.c_int(effect), ap[0], ap[1]))

    def live_wait(self):
        self._check(lib().rtdd_live_wait(self._h))

    def live_pending(self):
        return int(lib().rtdd_live_pending(self._h))

    def refine_depth(self, method=METHOD_RED_BLACK_GS, maxIterations=200000, tolerance=1e-4, checkEvery=0, relaxation=RELAXATION_AUTO):
        """rtdd_refine_depth: converge the finest level of the last estimate in place.  Returns (iterations, residual)."""
        params = SolveParams(method, maxIterations, tolerance, checkEvery, relaxation if method == METHOD_RED_BLACK_GS else 0.0)
        info = SolveInfo()
        self._check(lib().rtdd_refine_depth(self._h, C.byref(params), C.byref(info)))
        self.last_cycles = info.cycles
        return info.iterations, info.residual

    def bgr2gray(self, bgr, gray, rows, cols):
        b, bp = _img(bgr); g, gp = _img(gray)
        self._check(lib().rtdd_bgr2gray(self._h, b, bp, g, gp, C.c_int(rows), C.c_int(cols)))

    def pyrdown_gray(self, src, rows, cols, dst):
        s, sp = _img(src); d, dp = _img(dst)
        self._check(lib().rtdd_pyrdown_gray(self._h, s, sp, C.c_int(rows), C.c_int(cols), d, dp))

    def pyrup_depth(self, src, rows, cols, dst, drows, dcols):
        s, sp = _img(src); d, dp = _img(dst)
        self._check(lib().rtdd_pyrup_depth(self._h, s, sp, C.c_int(rows), C.c_int(cols), d, dp, C.c_int(drows), C.c_int(dcols)))

    def depth_to_u8(self, src, dst, rows, cols):
        s, sp = _img(src); d, dp = _img(dst)
        self._check(lib().rtdd_depth_to_u8(self._h, s, sp, d, dp, C.c_int(rows), C.c_int(cols)))


# ---- pitched device images (torch is plumbing for device memory only) ----------------------------
def device_image(host, device="cuda:0", align=512):
    """Upload a numpy image [rows, cols] or [rows, cols, 3] into a PITCHED device tensor whose row
    stride is padded to `align` bytes, like cudaMallocPitch / cv::cuda::GpuMat in the reference
    (/root/reference/src/main.cpp:117-137).  Returns the logical view (use .stride(0) for the pitch)."""
    import numpy as np
    import torch
    host = np.ascontiguousarray(host)
    rows = host.shape[0]
    row_elems = int(np.prod(host.shape[1:]))
    item = host.itemsize
    pitch_elems = ((row_elems * item + align - 1) // align * align) // item
    base = torch.zeros((rows, pitch_elems), dtype=torch.from_numpy(host[:0]).dtype, device=device)
    view = base[:, :row_elems]
    view.copy_(torch.from_numpy(host.reshape(rows, row_elems)).to(device))
    if host.ndim == 3:
        view = view.unflatten(1, host.shape[1:])
    return view


class host_image:
    """A numpy array over page-locked host memory (rtdd_host_alloc), so that live-mode copies are asynchronous.  `.a` is the array;
    free() or the garbage collector releases the memory (keep the object alive while frames that use it are in flight)."""

    def __init__(self, shape, dtype="uint8"):
        import numpy as np
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._p = C.c_void_p()
        rc = lib().rtdd_host_alloc(C.byref(self._p), C.c_size_t(n))
        if rc != RTDD_OK:
            raise RtddError(rc, "rtdd_host_alloc")
        self.a = np.frombuffer((C.c_char * n).from_address(self._p.value), dtype=dtype).reshape(shape)

    def free(self):
        if self._p is not None and self._p.value:
            self.a = None
            lib().rtdd_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def to_host(view):
    return view.cpu().contiguous().numpy()
