"""Builds libanirec.so (the gfx950 HIP kernels + C ABI) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU; the .so is git-ignored but travels
to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libanirec.so")
SOURCES = ["anirec_misc.hip", "anirec_train.hip", "anirec_infer.hip", "anirec_topk_mfma.hip", "anirec_predict_mfma.hip",
           "anirec_ingest.hip", "anirec_recs.hip"]


def hipcc_path() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libanirec.so cannot be built")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(PKG), "include", "anirec.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    objs = []
    cc = hipcc_path()
    os.makedirs(os.path.join(PKG, "_obj"), exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(PKG, "_obj", src.replace(".hip", ".o"))
        cmd = [cc, "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall",
               "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode(errors="replace"))
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
