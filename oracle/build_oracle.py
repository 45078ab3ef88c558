"""Builds the plain-C oracle (oracle/anirec_oracle.c) into oracle/_build/liborc.so with gcc.

The reference is pure Python with no compilable sources, so there is no oracle/_ref build
("Oracle").  Test infrastructure only.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "anirec_oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "liborc.so")


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) > os.path.getmtime(SRC):
        return LIB
    # -march=x86-64-v3 (AVX2/FMA-capable ISA, but contraction is OFF so no fused ops are formed
    # except the explicit fmaf() calls); not -march=native: the .so travels to the GPU box
    cmd = ["gcc", "-O3", "-march=x86-64-v3", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-shared",
           "-fPIC", SRC, "-o", LIB + ".tmp", "-lm"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("gcc failed:\n" + r.stdout.decode(errors="replace"))
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
