#!/usr/bin/env python3
"""Static VALU instruction count of the Jacobi sweep's hot loop, from the disassembly of the BUILT object.

bench.py prices the sweep's VALU roofline with "VALU operations per pixel-sweep" -- a count of the instructions the fast path of
k_sweep_blocked<32, 1024, 3, true, true> issues for one pair of sweeps over a thread's 12 pixels.  That number used to be a constant
typed into bench.py; this module recounts it from realtimedepthdiffusion_amd/csrc/sweep_blocked.o so that the constant cannot drift
away from the code (tests/test_isa_hazards.py asserts the two agree; scripts/make_counters_json.py refuses to write a record otherwise).

How the loop is found: every backward branch of the function closes a loop; from its target the instructions are walked along the
FALL-THROUGH path (conditional branches not taken -- the waiting loops, the full-divide path and the time-out reports all hang off
taken branches, by construction: __builtin_expect -- unconditional branches followed) until the walk reaches a branch back to the loop's
head.  The sweep pair is the loop whose fall-through path holds exactly 24 v_med3_f32 (12 pixels x 2 sweeps) and no v_div_scale_f32.
"""
import glob
import os
import re
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
HOT_KERNEL = "k_sweep_blockedILi32ELi1024ELi3ELb1ELb1"         # k_sweep_blocked<32, 1024, 3, true, true>: the 1080p persistent instantiation
PIXELS_PER_THREAD = 12


def disassemble(obj):
    """{mangled function name: [(address, instruction text, branch target address or None), ...]} of the gfx950 code object inside `obj`."""
    tmp = tempfile.mkdtemp(prefix="isa_count_")
    try:
        local = os.path.join(tmp, "x.o")
        shutil.copy(obj, local)
        subprocess.check_call([OBJDUMP, "--offloading", local], stdout=subprocess.DEVNULL, cwd=tmp)
        dev = sorted(glob.glob(local + ".*gfx950*"))
        if not dev:
            raise RuntimeError("no gfx950 code object in " + obj)
        text = "\n".join(subprocess.check_output([OBJDUMP, "-d", d], text=True) for d in dev)      # (a shared library holds one code object per translation unit)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    funcs, cur, base = {}, None, 0
    for line in text.split("\n"):
        m = re.match(r"^([0-9a-f]+) <([^>]+)>:", line)
        if m:
            base = int(m.group(1), 16); cur = funcs.setdefault(m.group(2), [])
            continue
        m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):\s+[0-9A-Fa-f ]+(?:<[^>+]+\+0x([0-9a-fA-F]+)>)?\s*$", line)
        if m and cur is not None:
            cur.append((int(m.group(2), 16), m.group(1).strip(), base + int(m.group(3), 16) if m.group(3) else None))
    return funcs


def fallthrough_loops(instrs):
    """For every backward branch: the instructions on the fall-through path from its target back to a branch to that target."""
    index = {a: i for i, (a, _, _) in enumerate(instrs)}
    heads = sorted({t for a, txt, t in instrs if t is not None and t <= a and txt.startswith(("s_cbranch", "s_branch")) and t in index})
    loops = []
    for head in heads:
        i, path, steps = index[head], [], 0
        while 0 <= i < len(instrs) and steps < 20000:
            a, txt, t = instrs[i]
            steps += 1
            if txt.startswith(("s_cbranch", "s_branch")) and t == head:
                loops.append((head, path))
                break
            if txt.startswith("s_endpgm"):
                break
            path.append(txt)
            if txt.startswith("s_branch") and t is not None:
                if t not in index:
                    break
                i = index[t]
                continue
            i += 1
    return loops


def sweep_pair(obj=None, kernel=HOT_KERNEL):
    """Instruction census of the fast sweep pair: {'valu', 'salu', 'lds', 'per_pixel_sweep', 'by_mnemonic'}."""
    if obj is None:                                # the object where it exists (a build tree), else the library itself (what travels to a GPU box)
        obj = os.path.join(ROOT, "realtimedepthdiffusion_amd", "csrc", "sweep_blocked.o")
        if not os.path.exists(obj):
            obj = os.path.join(ROOT, "realtimedepthdiffusion_amd", "librtdd.so")
    funcs = disassemble(obj)
    names = [n for n in funcs if kernel in n]
    if len(names) != 1:
        raise RuntimeError(f"{kernel}: {len(names)} functions match in {obj}")
    found = []
    for head, path in fallthrough_loops(funcs[names[0]]):
        med3 = sum(t.startswith("v_med3_f32") for t in path)
        if med3 == 2 * PIXELS_PER_THREAD and not any(t.startswith("v_div_scale") for t in path):
            found.append(path)
    if len(found) != 1:
        raise RuntimeError(f"{kernel}: expected ONE loop with 24 v_med3_f32 and no full divide on its fall-through path, found {len(found)}")
    path = found[0]
    by = {}
    for t in path:
        by[t.split()[0]] = by.get(t.split()[0], 0) + 1
    valu = sum(n for k, n in by.items() if k.startswith("v_"))
    return {"valu": valu, "salu": sum(n for k, n in by.items() if k.startswith("s_")), "lds": sum(n for k, n in by.items() if k.startswith("ds_")),
            "instructions": len(path), "per_pixel_sweep": valu / (2.0 * PIXELS_PER_THREAD), "by_mnemonic": dict(sorted(by.items(), key=lambda kv: -kv[1]))}


if __name__ == "__main__":
    import json
    c = sweep_pair()
    print(json.dumps(c, indent=1))
