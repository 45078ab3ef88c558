"""Thin torch-tensor wrappers over the inference / utility entry points of libanirec.

Names follow the reference: ``get_weights`` row-normalisation (similar_anime.py:136-171),
cosine neighbours (similar_users.py:290-296), ``model.predict`` (model_recs.py:394).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import DIM, MAX_TOPK


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(x, device):
    t = torch.as_tensor(x, device=device)
    return t.to(torch.float32).contiguous()


def _i32(x, device):
    return torch.as_tensor(x, device=device).to(torch.int32).contiguous()


def _need_gpu():
    if not torch.cuda.is_available():
        raise _lib.AnirecError("no GPU: the anime_recommendations_amd hot path needs an MI355X")


def rownorm(W, device="cuda:0"):
    """``W / np.linalg.norm(W, axis=1).reshape(-1, 1)`` on the GPU (fp32, no epsilon)."""
    _need_gpu()
    lib = _lib.load()
    W = _f32(W, device)
    assert W.dim() == 2 and W.shape[1] == DIM
    out = torch.empty_like(W)
    _lib.check(lib.anirec_rownorm(_lib.ptr(W), W.shape[0], _lib.ptr(out), _stream()), "anirec_rownorm")
    return out


def cosine_scores(What, q):
    """``np.dot(What, What[q])`` with the library's fixed fp32 summation order."""
    _need_gpu()
    lib = _lib.load()
    assert What.is_cuda and What.dtype == torch.float32 and What.shape[1] == DIM
    out = torch.empty(What.shape[0], dtype=torch.float32, device=What.device)
    _lib.check(lib.anirec_cosine_scores(_lib.ptr(What), What.shape[0], int(q), _lib.ptr(out), _stream()),
               "anirec_cosine_scores")
    return out


def cosine_topk(What, queries, k, exclude_self=True, keep=None, workspace=None):
    """Top-k rows by descending cosine for each query row index.

    Returns (idx int32 [nq,k], score fp32 [nq,k]); padded with -1 / NaN.
    Ties -> ascending row index; NaN scores rank last.
    """
    _need_gpu()
    lib = _lib.load()
    assert What.is_cuda and What.dtype == torch.float32 and What.shape[1] == DIM
    if not (1 <= k <= MAX_TOPK):
        raise ValueError("k must be in 1..%d" % MAX_TOPK)
    dev = What.device
    n = What.shape[0]
    if not isinstance(queries, torch.Tensor) or not queries.is_cuda:
        qh = np.asarray(queries if not isinstance(queries, torch.Tensor) else queries.numpy())
        if qh.size and (qh.min() < 0 or qh.max() >= n):       # validated on the host: no device sync
            raise ValueError("query row out of range")
        q = _i32(qh, dev)
    else:
        q = _i32(queries, dev)
        if q.numel() and bool(((q < 0) | (q >= n)).any()):     # one sync instead of two
            raise ValueError("query row out of range")
    nq = int(q.numel())
    out_i = torch.empty(nq, k, dtype=torch.int32, device=dev)
    out_s = torch.empty(nq, k, dtype=torch.float32, device=dev)
    if nq == 0:
        return out_i, out_s
    keep_t = None
    if keep is not None:
        keep_t = torch.as_tensor(keep, device=dev).to(torch.uint8).contiguous()
        assert keep_t.numel() == n
    if workspace is None:
        workspace = torch.empty(int(lib.anirec_topk_workspace_bytes(n, nq)), dtype=torch.uint8, device=dev)
    _lib.check(lib.anirec_cosine_topk(_lib.ptr(What), n, _lib.ptr(q), nq, _lib.ptr(keep_t),
                                      int(bool(exclude_self)), int(k), _lib.ptr(out_i), _lib.ptr(out_s),
                                      _lib.ptr(workspace), workspace.numel(), _stream()),
               "anirec_cosine_topk")
    return out_i, out_s


def topk_job_plan(nq, k, prior="auto", batch=None, lanes=None):
    """The library's default batch plan of a cosine_topk_mfma job: (starts [n_batches + 1], learn_batches, lanes)."""
    lib = _lib.load()
    st = (C.c_int32 * (_lib.TOPK_MAX_BATCHES + 1))()
    nb, nl = C.c_int32(0), C.c_int32(0)
    if lanes is None:
        lanes = int(os.environ.get("ANIREC_TOPK_LANES", "2"))
    lanes = max(1, min(4, int(lanes)))
    _lib.check(lib.anirec_cosine_topk_job_plan(int(nq), int(k), int(prior == "auto"), int(batch or 0), lanes, st,
                                               C.byref(nb), C.byref(nl)), "anirec_cosine_topk_job_plan")
    return [int(st[i]) for i in range(nb.value + 1)], int(nl.value), lanes


_JOB_WS = {}


def _job_workspace(nbytes, dev):
    """The job's workspace, kept between calls (grow-only, one per device and stream): at 350 k rows it is several GB
    (the all-pairs job: two inboxes of n x 256 x 8 B = 1.4 GB, two chains' logs of ~2 GB each, the candidate buffers —
    7-8 GB in all), and torch's caching allocator may carve a freed block of that size up for the next small
    allocations, so that the following job pays a hipMalloc inside its call.  Stream-ordered reuse is safe: every job
    runs on torch's current stream and ends joined to it.  ``release_workspaces`` drops the cache; a workspace larger
    than ANIREC_TOPK_WS_CACHE_GB (default 16) is never cached."""
    cap = float(os.environ.get("ANIREC_TOPK_WS_CACHE_GB", "16")) * (1 << 30)
    if nbytes > cap:
        return torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
    key = (torch.device(dev).index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _JOB_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        _JOB_WS.pop(key, None)
        ws = None
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        _JOB_WS[key] = ws
    return ws


def release_workspaces():
    """Give the cached job workspaces back to the allocator."""
    _JOB_WS.clear()


def _allpairs_pilot(What, n, k_eff, stats):
    """Is the table sparse enough for the all-pairs shortcut?  The shortcut hands a row every pair that reaches the
    job's PRIOR (a low quantile of the rows' k-th best scores) through a 256-entry inbox; a row with far more than k
    neighbours above the prior — dense clusters — overflows it and is re-run, and a table full of such rows is several
    times SLOWER than the plain job (measured: clusters of 1 000 rows at n = 200 k: 147 vs 21 ms; clusters of 600:
    23 vs 48 ms, the shortcut still ahead).  So: 512 sample rows against a strided sample of 32 768 rows (one small
    MFMA job), the sample's own estimate of the prior, and the share of sample rows with more than ~6 k neighbours
    above it; the shortcut is taken when that share is at most 2 %.  ~0.4 ms of a >= 10 ms job."""
    m = 32768
    dev = What.device
    idx = torch.arange(m, device=dev, dtype=torch.int64) * (n // m)
    S = What[idx].contiguous()
    k_sub = max(1, int(round(k_eff * m / float(n))))
    kk = min(MAX_TOPK - 1, 6 * k_sub + 16)
    _, sc, _ = cosine_topk_mfma(S, torch.arange(512, dtype=torch.int32, device=dev), kk, exclude_self=True, prior=None,
                                allpairs=False, lanes=1)
    kth = sc[:, k_sub - 1]
    kth = kth[kth == kth]
    if kth.numel() < 256:
        stats["pilot"] = {"sample_rows": int(kth.numel()), "dense_share": None, "allpairs": False}
        return False
    # (the 10 % quantile, not the job's 0.5 %: a sample row's estimate of its k-th best is the k_sub-th best of 32 768
    # keys, noisy by ~1 / sqrt(k_sub) in rank; the extreme quantile of 512 such estimates would sit far below the prior
    # the 16 384 exact learning rows produce)
    prior_est = torch.quantile(kth, 0.10) - 0.0101
    dense = float(((sc >= prior_est).sum(1) >= kk).float().mean())
    stats["pilot"] = {"prior_estimate": float(prior_est), "dense_share": dense, "allpairs": dense <= 0.02}
    return dense <= 0.02


def topk_allpairs_plan(n, k, lanes=None, main_batches=0):
    """The library's plan of the all-pairs job (learning batch + batches of equal work): (starts, learn_batches, lanes)."""
    lib = _lib.load()
    st = (C.c_int32 * (_lib.TOPK_MAX_BATCHES + 1))()
    nb, nl = C.c_int32(0), C.c_int32(0)
    if lanes is None:
        lanes = int(os.environ.get("ANIREC_TOPK_LANES", "2"))
    lanes = max(1, min(2, int(lanes)))
    _lib.check(lib.anirec_cosine_topk_allpairs_plan(int(n), int(k), lanes, int(main_batches), st, C.byref(nb), C.byref(nl)),
               "anirec_cosine_topk_allpairs_plan")
    return [int(st[i]) for i in range(nb.value + 1)], int(nl.value), lanes


def cosine_topk_mfma(What, queries, k, exclude_self=True, keep=None, batch=None, fallback=True, prior="auto",
                     cand_timing=None, lanes=None, stats=None, allpairs="auto"):
    """cosine_topk on the matrix cores (fp16 MFMA candidates + exact fp32 re-rank); rows the
    kernel could not prove complete are transparently re-run through the exact kernels.
    ``What`` must hold unit-norm rows (``rownorm`` output, as at every reference call site): the MFMA error
    window is proven for unit vectors; the kernel checks it and un-normalised input sends EVERY query to the
    exact path (correct, slow).
    The whole job is ONE library call (anirec_cosine_topk_job): keys converted once, the queries cut into batches
    (``topk_job_plan``; ``batch`` = most rows per batch) that run as ``lanes`` interleaved stream-ordered chains, so
    the per-row refresh / re-rank work of one batch runs beside the MFMA kernel of another.
    ``prior``: a threshold every row starts from instead of "below every cosine" (``None``), a float, or "auto":
    for 49 152 queries or more a first batch of 16 384 rows runs without a prior and the k-th best scores of its rows
    give one for the others — their 0.5 % quantile minus the error window and a margin, computed on the device —
    which spares those rows most of their ~k ln(n) early candidates.  A row whose own threshold lies below the prior
    comes out unproven and is re-run without one.  Results are identical either way.
    ``allpairs``: when every row is a query, in order (``queries`` == arange(n), no ``keep``), and the job learns a
    prior, a batch computes its dot products with the rows of later batches once for both sides (cosine is symmetric;
    include/anirec.h, prior_mode 3).  "auto" checks the query list on the device; ``False`` / env ``ANIREC_TOPK_SYM=0``
    keep every batch on the whole key stream.  Results are identical either way.
    ``cand_timing``: a dict that receives ``ms`` / ``launches`` of all MFMA candidate-kernel launches of this call
    (HIP events on their stream, every batch and re-run included; the job then runs on one chain and blocks per
    batch — bench.py only).  ``stats``: a dict that receives ``batches``, ``learn_batches``, ``lanes``, ``starts``,
    ``rerun_rows`` (unproven under the prior, run again without) and ``fallback_rows``.
    Returns (idx, score, n_fallback)."""
    _need_gpu()
    lib = _lib.load()
    assert What.is_cuda and What.dtype == torch.float32 and What.shape[1] == DIM
    if not (1 <= k <= MAX_TOPK - 1):
        raise ValueError("k must be in 1..%d" % (MAX_TOPK - 1))
    dev, n = What.device, What.shape[0]
    q = _i32(queries, dev)
    nq = int(q.numel())
    out_i = torch.empty(nq, k, dtype=torch.int32, device=dev)
    out_s = torch.empty(nq, k, dtype=torch.float32, device=dev)
    if stats is None:
        stats = {}
    stats.update(batches=0, learn_batches=0, rerun_rows=0, fallback_rows=0)
    if nq == 0:                     # an empty query shard (dist_infer on more ranks than queries)
        return out_i, out_s, 0
    keep_t = None
    if keep is not None:
        keep_t = torch.as_tensor(keep, device=dev).to(torch.uint8).contiguous()
    # (below 196 608 rows the all-pairs plan is the default plan and the shortcut buys nothing: no device check, no sync)
    want_sym = (bool(allpairs) and prior == "auto" and nq == n and keep_t is None
                and os.environ.get("ANIREC_TOPK_SYM", "1") != "0"
                and (allpairs is True or (n >= 196608
                                          and bool(torch.equal(q, torch.arange(n, dtype=torch.int32, device=dev))))))
    if want_sym and allpairs == "auto" and os.environ.get("ANIREC_TOPK_PILOT", "1") != "0":
        if batch is None:
            # size the cached workspace for the job that follows BEFORE the pilot's small job takes the slot (a cold
            # call would otherwise allocate the pilot's workspace, drop it and allocate the main one)
            st0, _, ln0 = topk_allpairs_plan(n, k, lanes)
            rows0 = max(st0[i + 1] - st0[i] for i in range(len(st0) - 1))
            _job_workspace(int(lib.anirec_cosine_topk_allpairs_workspace_bytes(n, rows0, max(1, min(ln0, 2)))), dev)
        want_sym = _allpairs_pilot(What, n, k + int(bool(exclude_self)), stats)
    if want_sym and batch is None:
        starts, learn, lanes = topk_allpairs_plan(n, k, lanes)
    else:
        starts, learn, lanes = topk_job_plan(nq, k, prior, batch, lanes)
    nb = len(starts) - 1
    if prior is None:
        mode, theta0 = 0, 0.0
    elif prior == "auto":
        mode, theta0 = (1 if learn else 0), 0.0
    else:
        mode, theta0 = 2, float(prior)
    rows = max(starts[i + 1] - starts[i] for i in range(nb))
    eff_lanes = max(1, min(lanes, nb - learn))
    sym = (want_sym and mode == 1 and nb >= 3 and eff_lanes <= 2 and all(x % 128 == 0 for x in starts[1:-1]))
    if sym:
        mode = 3
        ws_bytes = int(lib.anirec_cosine_topk_allpairs_workspace_bytes(n, rows, eff_lanes))
    else:
        ws_bytes = int(lib.anirec_cosine_topk_job_workspace_bytes(n, rows, eff_lanes))
    ws = _job_workspace(ws_bytes, dev)
    flags = torch.empty(nq, dtype=torch.int32, device=dev)
    st = (C.c_int32 * (nb + 1))(*starts)
    if cand_timing is not None:
        topk_mfma_timing(True)
    _lib.check(lib.anirec_cosine_topk_job(_lib.ptr(What), n, _lib.ptr(q), nq, _lib.ptr(keep_t), int(bool(exclude_self)),
                                          int(k), mode, theta0, st, nb, learn, eff_lanes, _lib.ptr(out_i), _lib.ptr(out_s),
                                          _lib.ptr(flags), _lib.ptr(ws), ws.numel(), _stream()), "anirec_cosine_topk_job")
    if cand_timing is not None:
        ms, nl = topk_mfma_timing(False)
        cand_timing["ms"] = cand_timing.get("ms", 0.0) + ms
        cand_timing["launches"] = cand_timing.get("launches", 0) + nl
    stats.update(batches=nb, learn_batches=learn, lanes=eff_lanes, starts=starts, allpairs=sym)
    bad = torch.nonzero(flags, as_tuple=False).flatten()       # the one host sync of the job
    n_fb = 0
    if bad.numel():     # diagnostics: how many rows carry each flag bit (1 overflow, 2 unproven, 4 not unit-norm)
        fb = flags[bad]
        stats["flag_rows"] = {bit: int(((fb & bit) != 0).sum()) for bit in (1, 2, 4)}
    if bad.numel() and mode != 0:
        # rows the prior was too high for (or otherwise unproven): once more without it
        sub = {}
        fi, fs, n_fb = cosine_topk_mfma(What, q[bad], k, exclude_self=exclude_self, keep=keep_t, batch=batch,
                                        fallback=fallback, prior=None, cand_timing=cand_timing, lanes=lanes, stats=sub)
        out_i[bad] = fi
        out_s[bad] = fs
        stats["rerun_rows"] = int(bad.numel())
    elif bad.numel():
        n_fb = int(bad.numel())
        if fallback:
            fi, fs = cosine_topk(What, q[bad], k, exclude_self=exclude_self, keep=keep_t)
            out_i[bad] = fi
            out_s[bad] = fs
    stats["fallback_rows"] = n_fb
    return out_i, out_s, n_fb


def topk_mfma_timing(enable):
    """Arm / disarm HIP-event timing of the MFMA candidate kernel; returns (ms, launches) summed over the calls
    made since it was last armed (bench.py's roofline leg)."""
    lib = _lib.load()
    ms, nl = C.c_float(0.0), C.c_int32(0)
    _lib.check(lib.anirec_topk_mfma_timing(int(bool(enable)), C.byref(ms), C.byref(nl)), "anirec_topk_mfma_timing")
    return float(ms.value), int(nl.value)


def _head_struct(head):
    return _lib.Head(float(head["w"]), float(head["b"]), float(head["gamma"]), float(head["beta"]),
                     float(head["mov_mean"]), float(head["mov_var"]))


def predict_pairs(U, A, head, user_idx, anime_idx):
    """``model.predict([user_arr, anime_arr]).flatten()`` (BN inference mode)."""
    _need_gpu()
    lib = _lib.load()
    dev = U.device
    ui, ai = _i32(user_idx, dev), _i32(anime_idx, dev)
    assert ui.numel() == ai.numel()
    p = torch.empty(ui.numel(), dtype=torch.float32, device=dev)
    h = _head_struct(head)
    _lib.check(lib.anirec_predict_pairs(_lib.ptr(U), _lib.ptr(A), _lib.ptr(ui), _lib.ptr(ai),
                                        int(ui.numel()), C.byref(h), _lib.ptr(p), _stream()),
               "anirec_predict_pairs")
    return p


def predict_grid(U, A, head, users):
    """Predicted rating of every anime for each listed user -> [len(users), n_anime] fp32."""
    _need_gpu()
    lib = _lib.load()
    dev = U.device
    us = _i32(users, dev)
    n_a, n_q = A.shape[0], int(us.numel())
    out = torch.empty(n_q, n_a, dtype=torch.float32, device=dev)
    ws = torch.empty(int(lib.anirec_predict_workspace_bytes(n_a, max(n_q, 1), 0)), dtype=torch.uint8, device=dev)
    h = _head_struct(head)
    _lib.check(lib.anirec_predict_grid(_lib.ptr(U), _lib.ptr(A), n_a, _lib.ptr(us), n_q, C.byref(h),
                                       _lib.ptr(out), _lib.ptr(ws), ws.numel(), _stream()),
               "anirec_predict_grid")
    return out


def predict_grid_mfma(U, A, head, users, out=None):
    """predict_grid on the matrix cores (split-fp16 MFMA, ratings within 1e-5 of the fp32 path)."""
    _need_gpu()
    lib = _lib.load()
    dev = U.device
    us = _i32(users, dev)
    n_a, n_q = A.shape[0], int(us.numel())
    if out is None:
        out = torch.empty(n_q, n_a, dtype=torch.float32, device=dev)
    ws = torch.empty(int(lib.anirec_predict_mfma_workspace_bytes(n_a, max(n_q, 1))), dtype=torch.uint8, device=dev)
    h = _head_struct(head)
    _lib.check(lib.anirec_predict_grid_mfma(_lib.ptr(U), _lib.ptr(A), n_a, _lib.ptr(us), n_q, C.byref(h),
                                            _lib.ptr(out), _lib.ptr(ws), ws.numel(), _stream()),
               "anirec_predict_grid_mfma")
    return out


def predict_topk(U, A, head, users, k, watched_bits=None):
    """Top-k unwatched anime by predicted rating per user.  watched_bits: uint32/int32
    [n_users, ceil(n_anime/32)] (bit set = watched) or None."""
    _need_gpu()
    lib = _lib.load()
    dev = U.device
    us = _i32(users, dev)
    n_a, n_q = A.shape[0], int(us.numel())
    out_i = torch.empty(n_q, k, dtype=torch.int32, device=dev)
    out_p = torch.empty(n_q, k, dtype=torch.float32, device=dev)
    if n_q == 0:
        return out_i, out_p
    wb = None
    if watched_bits is not None:
        wb = torch.as_tensor(watched_bits, device=dev).to(torch.int32).contiguous()
        assert wb.shape == (n_q, (n_a + 31) // 32)
    ws = torch.empty(int(lib.anirec_predict_workspace_bytes(n_a, n_q, 1)), dtype=torch.uint8, device=dev)
    h = _head_struct(head)
    _lib.check(lib.anirec_predict_topk(_lib.ptr(U), _lib.ptr(A), n_a, _lib.ptr(us), n_q, C.byref(h),
                                       _lib.ptr(wb), int(k), _lib.ptr(out_i), _lib.ptr(out_p),
                                       _lib.ptr(ws), ws.numel(), _stream()), "anirec_predict_topk")
    return out_i, out_p


def predict_topk_mfma(U, A, head, users, k, watched_bits=None, batch=131072, fallback=True):
    """predict_topk on the matrix cores (the batched model_recs path).  Users whose candidate window
    could not be proven complete are transparently re-run through the exact kernels.
    Returns (idx, p, n_fallback)."""
    _need_gpu()
    lib = _lib.load()
    dev = U.device
    us = _i32(users, dev)
    n_a, n_q = A.shape[0], int(us.numel())
    if not (1 <= k <= MAX_TOPK - 1):
        raise ValueError("k must be in 1..%d" % (MAX_TOPK - 1))
    out_i = torch.empty(n_q, k, dtype=torch.int32, device=dev)
    out_p = torch.empty(n_q, k, dtype=torch.float32, device=dev)
    if n_q == 0:
        return out_i, out_p, 0
    wb = None
    if watched_bits is not None:
        wb = torch.as_tensor(watched_bits, device=dev).to(torch.int32).contiguous()
        assert wb.shape == (n_q, (n_a + 31) // 32)
    h = _head_struct(head)
    bq = min(n_q, int(batch))
    ws = torch.empty(int(lib.anirec_predict_topk_mfma_workspace_bytes(n_a, bq)), dtype=torch.uint8, device=dev)
    flags = torch.empty(bq, dtype=torch.int32, device=dev)
    n_fb = 0
    for q0 in range(0, n_q, bq):
        cnt = min(bq, n_q - q0)
        wq = wb[q0:q0 + cnt] if wb is not None else None
        _lib.check(lib.anirec_predict_topk_mfma(_lib.ptr(U), _lib.ptr(A), n_a, _lib.ptr(us[q0:q0 + cnt]), cnt,
                                                C.byref(h), _lib.ptr(wq), int(k), _lib.ptr(out_i[q0:q0 + cnt]),
                                                _lib.ptr(out_p[q0:q0 + cnt]), _lib.ptr(flags), _lib.ptr(ws),
                                                ws.numel(), _stream()), "anirec_predict_topk_mfma")
        bad = torch.nonzero(flags[:cnt], as_tuple=False).flatten()
        n_fb += int(bad.numel())
        if bad.numel() and fallback:
            fi, fp = predict_topk(U, A, head, us[q0:q0 + cnt][bad], k, wq[bad] if wq is not None else None)
            out_i[q0 + bad] = fi
            out_p[q0 + bad] = fp
    return out_i, out_p, n_fb


def adam_flat(w, m, v, g, alpha):
    """In-place Keras-2.12 Adam dense update of flat fp32 tensors (bit-exact vs the oracle)."""
    _need_gpu()
    lib = _lib.load()
    _lib.check(lib.anirec_adam_flat(_lib.ptr(w), _lib.ptr(m), _lib.ptr(v), _lib.ptr(g), w.numel(),
                                    float(np.float32(alpha)), _stream()), "anirec_adam_flat")


def gather_ratings(user_idx, anime_idx, rating, perm):
    """Epoch shuffle: returns the three rating columns permuted by ``perm`` (int64)."""
    _need_gpu()
    lib = _lib.load()
    dev = user_idx.device
    perm = torch.as_tensor(perm, device=dev).to(torch.int64).contiguous()
    n = perm.numel()
    uo = torch.empty(n, dtype=torch.int32, device=dev)
    ao = torch.empty(n, dtype=torch.int32, device=dev)
    to = torch.empty(n, dtype=torch.float32, device=dev)
    _lib.check(lib.anirec_gather_ratings(_lib.ptr(user_idx), _lib.ptr(anime_idx), _lib.ptr(rating),
                                         _lib.ptr(perm), n, _lib.ptr(uo), _lib.ptr(ao), _lib.ptr(to),
                                         _stream()), "anirec_gather_ratings")
    return uo, ao, to
