#!/usr/bin/env python3
"""Derives the Itanium-mangled names of the reference's ten GPU* functions from the REFERENCE'S OWN headers
(/root/reference/include/GPUSolver.h, GPUImageProcessing.h, GPUDepthEffect.h, included in the order of src/main.cpp:9-11 --
two of them use size_t and rely on GPUSolver.h's <iostream> coming first) and writes tests/golden/reference_mangled_symbols.txt.
Run in the authoring container only; the fixture is data (symbol names), no reference source text."""
import os
import re
import subprocess
import sys

REF = "/root/reference/include"
NAMES = ["GPUAllocateDeviceMemory", "GPUFreeDeviceMemory", "GPULoadWeights", "GPUMatrixFreeSolver", "GPUConvertToFloat",
         "GPUPyrDownAnnotation", "GPUPaintImage", "GPUSimulateDefocus", "GPUSimulateDesaturation", "GPUSimulateHaze"]


def derive(include_dir=REF):
    src = '#include "GPUSolver.h"\n#include "GPUImageProcessing.h"\n#include "GPUDepthEffect.h"\nvoid *p[] = {%s};\n' % ", ".join("(void *)" + n for n in NAMES)
    asm = subprocess.check_output(["g++", "-x", "c++", "-I" + include_dir, "-S", "-o", "-", "-"], input=src, text=True)
    out = []
    for n in NAMES:
        m = re.findall(r"\b(_Z\d+%s\w*)" % n, asm)
        assert m, n
        out.append(m[0])
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    syms = derive()
    with open(os.path.join(here, "reference_mangled_symbols.txt"), "w") as f:
        f.write("\n".join(syms) + "\n")
    print("\n".join(syms))
