"""Local artifact store standing in for Weights & Biases (which the reference uses for every
inter-component hand-off: ``run.use_artifact(name:version, type).file()`` /
``run.log_artifact``, e.g. neural_network.py:35-39,237-275).  No network here, so artifacts
live under ``$ANIREC_ARTIFACT_DIR`` (default ``./artifacts``):

    <root>/<name>/v<N>/<file>     +  <root>/<name>/latest  (text file holding "v<N>")

``name[:version]`` resolves like W&B: ``:latest`` or no suffix -> newest, ``:v3`` -> that one.
"""
from __future__ import annotations

import json
import os
import shutil


def root_dir():
    return os.environ.get("ANIREC_ARTIFACT_DIR", os.path.join(os.getcwd(), "artifacts"))


def _split(spec):
    name, _, ver = spec.partition(":")
    return name, (ver or "latest")


def _versions(name):
    d = os.path.join(root_dir(), name)
    if not os.path.isdir(d):
        return []
    vs = [x for x in os.listdir(d) if x.startswith("v") and x[1:].isdigit()]
    return sorted(vs, key=lambda v: int(v[1:]))


def log_artifact(name, path, type=None, description=None, metadata=None):
    """Store ``path`` as the next version of artifact ``name``; returns the stored file path."""
    vs = _versions(name)
    ver = "v%d" % (int(vs[-1][1:]) + 1 if vs else 0)
    d = os.path.join(root_dir(), name, ver)
    os.makedirs(d, exist_ok=True)
    dst = os.path.join(d, os.path.basename(path))
    shutil.copyfile(path, dst)
    with open(os.path.join(d, "artifact.json"), "w") as f:
        json.dump({"name": name, "version": ver, "type": type, "description": description,
                   "metadata": metadata or {}, "file": os.path.basename(path)}, f, indent=1, default=str)
    with open(os.path.join(root_dir(), name, "latest"), "w") as f:
        f.write(ver)
    return dst


def use_artifact(spec, type=None):
    """Path of the single file of artifact ``name[:version]``.  A plain existing file path is
    accepted too (so components can be pointed at files directly)."""
    if os.path.isfile(spec):
        return spec
    name, ver = _split(spec)
    vs = _versions(name)
    if not vs:
        raise FileNotFoundError("artifact %r not found under %s" % (name, root_dir()))
    if ver == "latest":
        ver = vs[-1]
    if ver not in vs:
        raise FileNotFoundError("artifact %r has no version %s (have %s)" % (name, ver, vs))
    d = os.path.join(root_dir(), name, ver)
    meta = json.load(open(os.path.join(d, "artifact.json")))
    return os.path.join(d, meta["file"])
