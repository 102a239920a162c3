"""ctypes loader for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from realtimedepthdiffusion_amd/.  See the header of
rtdd_oracle.c for what pins it (the reference has no golden vectors: "parity unpinned").

All image arguments are numpy arrays whose last axis is contiguous; the row pitch handed to
C is ``arr.strides[0]`` so pitched (padded-row) views work exactly like the reference's
``GpuMat.ptr()/.step`` pairs.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("rtdd_oracle.c", "rtdd_cascade_oracle.c", "Makefile")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_residual.restype = C.c_float
        _lib.orc_residual_mt.restype = C.c_float
        _lib.orc_solve.restype = C.c_int
        _lib.orc_max_threads.restype = C.c_int
    return _lib


def _p(a):
    return C.c_void_p(a.ctypes.data)


def _pitch(a):
    if a.ndim == 2:
        assert a.strides[1] == a.itemsize, "last axis must be contiguous"
    else:
        assert a.strides[2] == a.itemsize and a.strides[1] == a.itemsize * a.shape[2], "pixels must be interleaved"
    return C.c_size_t(a.strides[0])


def max_threads():
    """Threads worth using here: OpenMP's count, capped by the cgroup CPU quota and the affinity mask
    (a GPU box gives one job a 16-core share of a 128-core host; oversubscribed OpenMP barriers crawl)."""
    n = int(lib().orc_max_threads())
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("RTDD_ORACLE_THREADS", "16"))))


def load_weights(beta):
    lut = np.empty(257, np.float32)
    lib().orc_load_weights(C.c_float(beta), _p(lut))
    return lut


def omega_schedule(n):
    om = np.empty(max(n, 1), np.float32)
    lib().orc_omega_schedule(C.c_int(n), _p(om))
    return om[:n]


def index_to_weight(gray, depth, level, max_level):
    rows, cols = gray.shape
    idx = np.empty((rows, cols, 2), np.int32)
    if depth is None:
        depth = np.zeros((rows, cols), np.float32)
    lib().orc_index_to_weight(_p(gray), _pitch(gray), _p(depth), _pitch(depth), _p(idx),
                              C.c_int(level), C.c_int(max_level), C.c_int(rows), C.c_int(cols))
    return idx


def sweep(x, idx, mask, prev, omega, lut, contract, gamma=np.float32(0.99), threads=1):
    """One K1 sweep on dense arrays; returns out (prev is updated in place).

    Positions with mask==255 are left as in ``x`` (the reference's ping-pong buffers both
    hold the input there)."""
    rows, cols = x.shape
    x = np.ascontiguousarray(x, np.float32)
    out = x.copy()
    lib().orc_sweep(_p(x), _p(idx), _p(mask), _pitch(mask), C.c_int(rows), C.c_int(cols), _p(out), _p(prev),
                    C.c_float(omega), C.c_float(gamma), _p(lut), C.c_int(contract), C.c_int(threads))
    return out


def solve(depth, mask, gray, max_iterations, level, max_level, lut, contract, threads=1, rows=None, cols=None):
    """GPUMatrixFreeSolver restatement; ``depth`` is updated in place (and returned)."""
    r = depth.shape[0] if rows is None else rows
    c = depth.shape[1] if cols is None else cols
    rc = lib().orc_solve(_p(depth), _pitch(depth), _p(mask), _pitch(mask), _p(gray), _pitch(gray),
                         C.c_int(r), C.c_int(c), C.c_int(max_iterations), C.c_int(level), C.c_int(max_level),
                         _p(lut), C.c_int(contract), C.c_int(threads))
    if rc != 0:
        raise MemoryError("orc_solve")
    return depth


def convert_to_float(src, dst, mask):
    rows, cols = mask.shape
    lib().orc_convert_to_float(_p(src), _pitch(src), _p(dst), _pitch(dst), _p(mask), _pitch(mask), C.c_int(rows), C.c_int(cols))
    return dst


def pyrdown_annotation(prev_scribble, prev_edited, curr_scribble, curr_edited):
    pr, pc = prev_scribble.shape
    cr, cc = curr_scribble.shape
    lib().orc_pyrdown_annotation(_p(prev_scribble), _pitch(prev_scribble), _p(prev_edited), _pitch(prev_edited),
                                 C.c_int(pr), C.c_int(pc), _p(curr_scribble), _pitch(curr_scribble),
                                 _p(curr_edited), _pitch(curr_edited), C.c_int(cr), C.c_int(cc))


def paint_image(x, y, color, radius, edited, scribble):
    rows, cols = scribble.shape
    lib().orc_paint_image(C.c_int(x), C.c_int(y), C.c_int(color), C.c_int(radius), _p(edited), _pitch(edited),
                          _p(scribble), _pitch(scribble), C.c_int(rows), C.c_int(cols))


def desaturate(orig, gray, depth, contract):
    rows, cols = gray.shape
    art = np.zeros_like(orig)
    lib().orc_desaturate(_p(orig), _pitch(orig), _p(gray), _pitch(gray), _p(depth), _pitch(depth), _p(art), _pitch(art),
                         C.c_int(rows), C.c_int(cols), C.c_int(contract))
    return art


def defocus(orig, depth, threads=1):
    rows, cols = depth.shape
    art = np.zeros_like(orig)
    lib().orc_defocus(_p(orig), _pitch(orig), _p(depth), _pitch(depth), _p(art), _pitch(art),
                      C.c_int(rows), C.c_int(cols), C.c_int(threads))
    return art


def defocus_at(orig, depth, ys, xs):
    """The literal gather of orc_defocus for the listed pixels only; returns [n, 3]."""
    rows, cols = depth.shape
    ys = np.ascontiguousarray(ys, np.int32); xs = np.ascontiguousarray(xs, np.int32)
    out = np.zeros((len(ys), 3), np.uint8)
    lib().orc_defocus_at(_p(orig), _pitch(orig), _p(depth), _pitch(depth), C.c_int(rows), C.c_int(cols), _p(ys), _p(xs), C.c_int(len(ys)), _p(out))
    return out


def haze(orig, depth, contract):
    rows, cols = depth.shape
    art = np.zeros_like(orig)
    lib().orc_haze(_p(orig), _pitch(orig), _p(depth), _pitch(depth), _p(art), _pitch(art),
                   C.c_int(rows), C.c_int(cols), C.c_int(contract))
    return art


def expf_det(x):
    """The deterministic exp of orc_haze (orc_expf_det) on an array; returns (values, how many differ from this host's libm expf)."""
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    lib().orc_expf_vs_libm.restype = C.c_int
    n = int(lib().orc_expf_vs_libm(_p(x), _p(out), C.c_int(x.size)))
    return out, n


def residual(x, idx, mask, lut, contract):
    rows, cols = x.shape
    x = np.ascontiguousarray(x, np.float32)
    return float(lib().orc_residual(_p(x), _p(idx), _p(mask), _pitch(mask), C.c_int(rows), C.c_int(cols), _p(lut), C.c_int(contract)))


def residual_mt(x, idx, mask, lut, contract, threads=None):
    """orc_residual on several cores (bit-identical: a maximum)."""
    rows, cols = x.shape
    x = np.ascontiguousarray(x, np.float32)
    return float(lib().orc_residual_mt(_p(x), _p(idx), _p(mask), _pitch(mask), C.c_int(rows), C.c_int(cols), _p(lut), C.c_int(contract),
                                       C.c_int(threads or max_threads())))


def rbgs_sweeps_mt(x, idx, mask, lut, contract, omega, nsweeps, threads=None):
    """nsweeps red-black sweeps in place, each colour's rows over several cores (bit-identical to rbgs_sweep)."""
    rows, cols = x.shape
    assert x.flags.c_contiguous and x.dtype == np.float32
    lib().orc_rbgs_sweeps_mt(_p(x), _p(idx), _p(mask), _pitch(mask), C.c_int(rows), C.c_int(cols), _p(lut), C.c_int(contract), C.c_float(omega),
                             C.c_int(nsweeps), C.c_int(threads or max_threads()))
    return x


def rbgs_sweep(x, idx, mask, lut, contract, omega=1.0):
    rows, cols = x.shape
    assert x.flags.c_contiguous and x.dtype == np.float32
    lib().orc_rbgs_sweep(_p(x), _p(idx), _p(mask), _pitch(mask), C.c_int(rows), C.c_int(cols), _p(lut), C.c_int(contract), C.c_float(omega))
    return x


def mg_solve(x, idx, mask, lut, contract, max_cycles, tolerance=0.0, check_every=1, alternative_seconds=0.0, cycle_seconds=0.0):
    """Multigrid V-cycles (extension; rtdd_mg_oracle.c) in place on x.  Returns (cycles, residual, levels)."""
    rows, cols = x.shape
    assert x.flags.c_contiguous and x.dtype == np.float32
    cyc = C.c_int(0); res = C.c_float(0)
    nlev = lib().orc_mg_solve(_p(x), _p(idx), _p(mask), _pitch(mask), C.c_int(rows), C.c_int(cols), _p(lut), C.c_int(contract),
                              C.c_int(max_cycles), C.c_float(tolerance), C.c_int(check_every), C.c_double(alternative_seconds), C.c_double(cycle_seconds),
                              C.byref(cyc), C.byref(res))
    return cyc.value, res.value, nlev


def mg_level(level, which):
    """Plane `which` (0-4 E,S,SE,SW,D; 5-8 P; 9-11 e,b,r) of level `level` of the last hierarchy mg_solve built."""
    r = C.c_int(0); c = C.c_int(0)
    if lib().orc_mg_level(C.c_int(level), C.c_int(which), None, C.byref(r), C.byref(c)) != 0:
        raise IndexError((level, which))
    out = np.zeros((r.value, c.value), np.float32)
    lib().orc_mg_level(C.c_int(level), C.c_int(which), _p(out), C.byref(r), C.byref(c))
    return out


# ---- third-party (OpenCV) restatements used by the cascade harness --------------------------
def bgr2gray(bgr):
    rows, cols = bgr.shape[:2]
    g = np.empty((rows, cols), np.uint8)
    lib().orc_bgr2gray(_p(bgr), _pitch(bgr), _p(g), _pitch(g), C.c_int(rows), C.c_int(cols))
    return g


def pyrdown_u8(src):
    rows, cols = src.shape
    d = np.empty(((rows + 1) // 2, (cols + 1) // 2), np.uint8)
    lib().orc_pyrdown_u8(_p(src), _pitch(src), C.c_int(rows), C.c_int(cols), _p(d), _pitch(d))
    return d


def pyrup_f32(src, drows, dcols, rows=None, cols=None, contract=1):
    """f32 pyrUp as src/main.cpp:272-279 calls it: cv::cuda::pyrUp when (drows, dcols) is exactly twice the source, else
    the host's cv::pyrUp with the explicit size (rtdd_cascade_oracle.c).  `contract`: nvcc's fmad in the CUDA branch."""
    r = src.shape[0] if rows is None else rows
    c = src.shape[1] if cols is None else cols
    d = np.empty((drows, dcols), np.float32)
    lib().orc_pyrup_f32(_p(src), _pitch(src), C.c_int(r), C.c_int(c), _p(d), _pitch(d), C.c_int(drows), C.c_int(dcols), C.c_int(contract))
    return d


def depth_to_u8(src):
    rows, cols = src.shape
    d = np.empty((rows, cols), np.uint8)
    lib().orc_depth_to_u8(_p(src), _pitch(src), _p(d), _pitch(d), C.c_int(rows), C.c_int(cols))
    return d
