"""Favourites and user-based recommendations — host side of ``anirec_user_favourites`` / ``anirec_user_recs``
(the consumer of the similar-users top-k).

Reference: user_recs/user_recs.py:377-404 (``fave_genres``: favourites = ratings at or above the 80th
percentile of the user's own ratings), :708-760 (``similar_user_recs``: count the similar users'
favourites the query user has not favourited, rank by count), similar_users.py:203-256.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _aligned(t):
    """Contiguous and 16-byte aligned (the kernels read four rows per lane): a sliced view is copied."""
    t = t.contiguous()
    return t if t.data_ptr() % 16 == 0 else t.clone()


def user_favourites(user_idx, anime_idx, rating, n_users, n_anime, percentile=80.0):
    """Returns (fav_bits int32 [n_users, ceil(n_anime/32)], threshold float64 [n_users]).
    Bit a of row u is set iff rating(u, a) >= np.percentile(ratings of u, percentile)."""
    if not torch.cuda.is_available():
        raise _lib.AnirecError("no GPU: the anime_recommendations_amd recs path needs an MI355X")
    lib = _lib.load()
    dev = user_idx.device
    u = _aligned(user_idx.to(torch.int32))
    a = _aligned(anime_idx.to(torch.int32))
    r = _aligned(rating.to(torch.float64))
    n = int(u.numel())
    ww = (int(n_anime) + 31) // 32
    fav = torch.empty(int(n_users), ww, dtype=torch.int32, device=dev)
    thr = torch.empty(int(n_users), dtype=torch.float64, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = torch.empty(int(lib.anirec_fav_workspace_bytes(n, int(n_users))), dtype=torch.uint8, device=dev)
    _lib.check(lib.anirec_user_favourites(_lib.ptr(u), _lib.ptr(a), _lib.ptr(r), n, int(n_users), int(n_anime),
                                          float(percentile), _lib.ptr(fav), _lib.ptr(thr), _lib.ptr(err),
                                          _lib.ptr(ws), ws.numel(), _stream()), "anirec_user_favourites")
    if int(err.item()):
        raise ValueError("user / anime index out of range")
    return fav, thr


def user_recs(fav_bits, n_anime, query_users, sim_users, n_recs):
    """Per query user the ``n_recs`` anime its similar users favourited most often (own favourites skipped).
    ``sim_users``: [nq, k_sim] user indices, best first, -1 = empty.  Returns (anime int32 [nq, n_recs]
    (-1 padded), count int32 [nq, n_recs])."""
    lib = _lib.load()
    dev = fav_bits.device
    q = torch.as_tensor(query_users, device=dev).to(torch.int32).contiguous()
    sim = torch.as_tensor(sim_users, device=dev).to(torch.int32).contiguous()
    nq, k_sim = int(sim.shape[0]), int(sim.shape[1])
    assert q.numel() == nq
    out_a = torch.empty(nq, int(n_recs), dtype=torch.int32, device=dev)
    out_c = torch.empty(nq, int(n_recs), dtype=torch.int32, device=dev)
    _lib.check(lib.anirec_user_recs(_lib.ptr(fav_bits), int(fav_bits.shape[0]), int(n_anime), _lib.ptr(q),
                                    _lib.ptr(sim), nq, k_sim, int(n_recs), _lib.ptr(out_a), _lib.ptr(out_c),
                                    _stream()), "anirec_user_recs")
    return out_a, out_c
