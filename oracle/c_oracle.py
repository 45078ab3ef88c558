"""ctypes wrapper of the plain-C oracle (oracle/anirec_oracle.c).  TEST INFRASTRUCTURE ONLY:
used by tests/ as the exact fmaf-chain checker and by bench.py's cpu_baseline ("port") leg."""
import ctypes as C

import numpy as np

from . import build_oracle


class Head(C.Structure):
    _fields_ = [("w", C.c_float), ("b", C.c_float), ("gamma", C.c_float), ("beta", C.c_float),
                ("m", C.c_float * 4), ("v", C.c_float * 4), ("mov_mean", C.c_float), ("mov_var", C.c_float)]


class Metrics(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("loss", "bce", "reg", "mse", "mu", "var")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_oracle.build())
        _lib.orc_max_threads.restype = C.c_int
        _lib.orc_train_run.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def max_threads():
    return int(lib().orc_max_threads())


def head_from(d):
    h = Head()
    h.w, h.b, h.gamma, h.beta = float(d["w"]), float(d["b"]), float(d["gamma"]), float(d["beta"])
    h.mov_mean, h.mov_var = float(d["mov_mean"]), float(d["mov_var"])
    for k in range(4):
        h.m[k] = float(d["m"][k]) if "m" in d else 0.0
        h.v[k] = float(d["v"][k]) if "v" in d else 0.0
    return h


def train_run(state, ui, ai, t, B, alphas, l2=1e-4):
    """Runs len(alphas) steps in place on an oracle state dict (same layout as anirec_oracle.new_state)."""
    for k in ("U", "A", "mU", "vU", "mA", "vA"):
        assert state[k].dtype == np.float32 and state[k].flags.c_contiguous
    ui = np.ascontiguousarray(ui, np.int32)
    ai = np.ascontiguousarray(ai, np.int32)
    t = np.ascontiguousarray(t, np.float32)
    alphas = np.ascontiguousarray(alphas, np.float32)
    h = head_from(state["head"])
    met = Metrics()
    rc = lib().orc_train_run(_p(state["U"]), _p(state["A"]), _p(state["mU"]), _p(state["vU"]),
                             _p(state["mA"]), _p(state["vA"]), C.c_int(state["U"].shape[0]),
                             C.c_int(state["A"].shape[0]), C.byref(h), _p(ui), _p(ai), _p(t),
                             C.c_int(len(ui)), C.c_int(B), _p(alphas), C.c_int(len(alphas)),
                             C.c_float(l2), C.byref(met))
    assert rc == 0
    hd = state["head"]
    hd["w"], hd["b"], hd["gamma"], hd["beta"] = (np.float32(x) for x in (h.w, h.b, h.gamma, h.beta))
    hd["mov_mean"], hd["mov_var"] = np.float32(h.mov_mean), np.float32(h.mov_var)
    hd["m"] = np.array(list(h.m), np.float32)
    hd["v"] = np.array(list(h.v), np.float32)
    state["t"] += len(alphas)
    return {k: getattr(met, k) for k in ("loss", "bce", "reg", "mse", "mu", "var")}


def cosine_scores(Wh, q):
    Wh = np.ascontiguousarray(Wh, np.float32)
    qv = np.ascontiguousarray(Wh[q], np.float32)
    out = np.empty(Wh.shape[0], np.float32)
    lib().orc_cosine_scores(_p(Wh), C.c_int(Wh.shape[0]), _p(qv), _p(out))
    return out


def rownorm(W):
    W = np.ascontiguousarray(W, np.float32)
    out = np.empty_like(W)
    lib().orc_rownorm(_p(W), C.c_int(W.shape[0]), _p(out))
    return out


def cosine_topk(Wh, queries, k, exclude_self=True, keep=None):
    Wh = np.ascontiguousarray(Wh, np.float32)
    q = np.ascontiguousarray(queries, np.int32)
    oi = np.empty((len(q), k), np.int32)
    ov = np.empty((len(q), k), np.float32)
    kp = None if keep is None else _p(np.ascontiguousarray(keep, np.uint8))
    lib().orc_cosine_topk(_p(Wh), C.c_int(Wh.shape[0]), _p(q), C.c_int(len(q)), C.c_int(k),
                          C.c_int(int(exclude_self)), kp, _p(oi), _p(ov))
    return oi, ov


def predict_grid(U, A, head, users):
    U = np.ascontiguousarray(U, np.float32)
    A = np.ascontiguousarray(A, np.float32)
    us = np.ascontiguousarray(users, np.int32)
    out = np.empty((len(us), A.shape[0]), np.float32)
    h = head_from(head)
    lib().orc_predict_grid(_p(U), _p(A), C.c_int(A.shape[0]), _p(us), C.c_int(len(us)), C.byref(h), _p(out))
    return out
