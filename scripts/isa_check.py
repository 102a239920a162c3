#!/usr/bin/env python3
"""Structural check of the no-acquire hand-off of the persistent kernels, on the disassembly of the BUILT objects.

The persistent hand-off reads its halo with `global_load_dwordx4 ... sc1` straight into the tile's registers and has NO agent-scope
acquire (csrc/persist_sync.hpp, exchange_wait<false>): the loaded registers are only valid behind the one `s_waitcnt vmcnt(0)` that
follows the loads, and the loads sit inside divergent `if`s.  A register copy, a spill (scratch_store, v_accvgpr_write) or any other use
the compiler placed between a load and that wait would silently read stale bits -- a wrong depth map, not an error.  So, in every
persistent instantiation: on EVERY path from an sc1 load (fall-through and branch targets alike) the first instruction that names one
of the load's destination registers comes behind an `s_waitcnt vmcnt(0)`.

realtimedepthdiffusion_amd.build() runs this after every build and falls back to the -DRTDD_EXCHANGE_ACQUIRE=1 hand-off when it
fails or cannot run (no llvm-objdump); tests/test_isa_hazards.py runs it too.  (the check used to be a linear scan in a test
that is skipped without the LLVM tools, so a compiler bump could have changed the code under it unnoticed.)"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_count

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _vgprs(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out |= set(range(int(a), int(b) + 1))
    out |= {int(n) for n in re.findall(r"\bv(\d+)\b", text)}
    return out


def check_object(obj, kernel):
    """(instantiations with sc1 loads, sc1 loads checked).  Raises AssertionError with the offending instruction."""
    funcs = isa_count.disassemble(obj)
    kernels = loads = 0
    for name, instrs in funcs.items():
        if kernel not in name:
            continue
        index = {a: i for i, (a, _, _) in enumerate(instrs)}
        here = 0
        for i, (addr, ins, _) in enumerate(instrs):
            if not (ins.startswith("global_load_dwordx4") and ins.rstrip().endswith("sc1")):
                continue
            dst = _vgprs(ins.split(None, 1)[1].split(",")[0])
            assert len(dst) == 4, ins
            # every path from the load: depth-first over fall-through and branch targets until a vmcnt(0) wait ends the path
            todo, seen = [i + 1], set()
            while todo:
                j = todo.pop()
                while j < len(instrs) and j not in seen:
                    seen.add(j)
                    a, later, target = instrs[j]
                    if later.startswith("s_waitcnt") and "vmcnt(0)" in later:
                        break
                    assert not later.startswith("s_endpgm"), f"{name}: a path from '{ins}' ends without s_waitcnt vmcnt(0)"
                    ops = later.split(None, 1)[1] if " " in later else ""
                    assert not (_vgprs(ops) & dst), f"{name}: '{later}' touches the registers of '{ins}' before the wait"
                    if later.startswith(("s_cbranch", "s_branch")) and target is not None:
                        assert target in index, f"{name}: branch out of the function behind '{ins}'"
                        todo.append(index[target])
                        if later.startswith("s_branch"):
                            break
                    j += 1
                assert len(seen) < 20000, name
            here += 1
        if here:
            kernels += 1
        loads += here
    return kernels, loads


def check_build(csrc=None):
    """The two objects of the product build: every persistent instantiation of k_sweep_blocked (26) and of k_rbgs_blocked (>= 4)."""
    csrc = csrc or os.path.join(ROOT, "realtimedepthdiffusion_amd", "csrc")
    k1, l1 = check_object(os.path.join(csrc, "sweep_blocked.o"), "k_sweep_blocked")
    assert k1 == 26 and l1 >= 52, f"k_sweep_blocked: {l1} sc1 loads in {k1} instantiations: was the hand-off rewritten?"
    k2, l2 = check_object(os.path.join(csrc, "rbgs_blocked.o"), "k_rbgs_blocked")
    assert k2 >= 4 and l2 >= 16, f"k_rbgs_blocked: {l2} sc1 loads in {k2} instantiations"
    return {"k_sweep_blocked": (k1, l1), "k_rbgs_blocked": (k2, l2)}


if __name__ == "__main__":
    print(check_build())
