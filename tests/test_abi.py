"""CPU-side checks of the drop-in boundary: librtdd.so builds for gfx950, loads without a GPU,
exports every symbol include/rtdd.h declares plus the reference's ten mangled C++ symbols, and
fails LOUDLY (no CPU fallback) when no device is present."""
import ctypes as C
import os
import re
import subprocess

import pytest

import realtimedepthdiffusion_amd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def so():
    return rt.build()


def _exported(so):
    out = subprocess.check_output(["nm", "-D", "--defined-only", so], text=True)
    return {line.split()[-1] for line in out.splitlines() if " T " in line}


def test_header_symbols_are_exported(so):
    header = open(os.path.join(ROOT, "include", "rtdd.h")).read()
    declared = set(re.findall(r"\b(rtdd_[a-z_0-9]+)\s*\(", header))
    assert declared == set(rt.C_ABI_SYMBOLS), declared ^ set(rt.C_ABI_SYMBOLS)
    exported = _exported(so)
    assert declared <= exported, declared - exported


def test_reference_mangled_symbols_are_exported(so):
    assert set(rt.DROPIN_SYMBOLS) <= _exported(so)
    # the one extern "C" name include/rtdd_dropin.hpp declares beside the ten: the shim's context, unmangled
    dropin = open(os.path.join(ROOT, "include", "rtdd_dropin.hpp")).read()
    assert set(re.findall(r'extern "C"[^;]*?\b(rtdd_[a-z_0-9]+)\s*\(', dropin)) == {"rtdd_dropin_context"}
    assert "rtdd_dropin_context" in _exported(so)
    # the mangled names really are what the reference's declarations produce
    hdr = os.path.join(ROOT, "include", "rtdd_dropin.hpp")
    src = '#include "%s"\nvoid* p[] = {(void*)GPUAllocateDeviceMemory,(void*)GPUFreeDeviceMemory,(void*)GPULoadWeights,(void*)GPUMatrixFreeSolver,(void*)GPUConvertToFloat,(void*)GPUPyrDownAnnotation,(void*)GPUPaintImage,(void*)GPUSimulateDefocus,(void*)GPUSimulateDesaturation,(void*)GPUSimulateHaze};' % hdr
    asm = subprocess.check_output(["g++", "-x", "c++", "-S", "-o", "-", "-"], input=src, text=True)
    for s in rt.DROPIN_SYMBOLS:
        assert s in asm


def test_mangled_names_come_from_the_references_own_headers(so):
    """tests/golden/reference_mangled_symbols.txt was derived by g++ from /root/reference/include/*.h (make_mangled.py): the
    library exports exactly those, the Python list agrees, and where the reference is present the derivation is repeated."""
    fixture = open(os.path.join(ROOT, "tests", "golden", "reference_mangled_symbols.txt")).read().split()
    assert fixture == rt.DROPIN_SYMBOLS and set(fixture) <= _exported(so)
    if os.path.isdir("/root/reference/include"):
        import importlib.util
        spec = importlib.util.spec_from_file_location("make_mangled", os.path.join(ROOT, "tests", "golden", "make_mangled.py"))
        mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
        assert mod.derive() == fixture


def test_no_signature_leaks_cxx_or_torch_types():
    header = open(os.path.join(ROOT, "include", "rtdd.h")).read()
    assert 'extern "C"' in header
    code = re.sub(r"/\*.*?\*/", "", header, flags=re.S)          # declarations only, comments stripped
    for banned in ("std::", "torch", "at::", "hipStream_t", "template", "class ", "&"):
        assert banned not in code, banned


def test_library_loads_and_reports_status_strings(so):
    L = rt.lib()
    assert L.rtdd_version() == 230          # 2xx: rtdd_solve_info is 36 bytes; 210: self-healing time-out; 220: re-armed persistence, batched estimates; 230: rtdd_pyramid_level_info (include/rtdd.h)
    assert L.rtdd_status_string(0) == b"ok"
    assert b"no CPU fallback" in L.rtdd_status_string(5)
    assert L.rtdd_ctx_create(C.c_int(0), None) == 1          # null out pointer -> RTDD_ERR_INVALID, no crash


def test_fails_loudly_without_a_device(so):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RtddError) as e:
        rt.Context(0)
    assert e.value.status == 5 and "no CPU fallback" in str(e.value)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "realtimedepthdiffusion_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle", text, re.M), f
                assert "liboracle" not in text and "orc_" not in text, f


def test_header_is_plain_c99_and_a_c_program_links(so, tmp_path):
    """include/rtdd.h must be usable from C (the reference's FFI surface is C-compatible): compile a C99 user with
    -pedantic -Werror and link it against librtdd.so (it is RUN on the GPU box by tests/test_gpu_harness.py)."""
    src = os.path.join(ROOT, "tests", "c_abi_smoke.c")
    exe = str(tmp_path / "c_abi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-o", exe, src,
                           "-L" + os.path.dirname(so), "-lrtdd", "-Wl,-rpath," + os.path.dirname(so)])
    assert os.path.exists(exe)
