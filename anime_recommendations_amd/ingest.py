"""Rating-table ingest on the GPU — host side of ``anirec_ingest_*``.

Mirrors the reference's preprocess step (preprocess/preprocess.py:13-40 ``drop_useless``, :52-105
``drop_half_watched``, :108-117 ``scale_ratings``) and the id encoding of ``get_df``
(neural_network/neural_network.py:41-60).  The columns live in HBM as int32 / float64 tensors; the
surviving rows keep their order and every value is bit-identical to what pandas produces.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib

NULL_I32 = -2 ** 31
COLUMNS = ("user_id", "anime_id", "rating", "watching_status", "watched_episodes")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu():
    if not torch.cuda.is_available():
        raise _lib.AnirecError("no GPU: the anime_recommendations_amd ingest path needs an MI355X")


class Columns(dict):
    """Device columns + ``bounds``: exclusive upper bounds of the id columns (the sizes of the direct-index
    tables), known without another pass over the column."""

    def __init__(self, *a, bounds=None, **kw):
        super().__init__(*a, **kw)
        self.bounds = dict(bounds or {})


def frame_to_columns(df, device="cuda:0"):
    """DataFrame (reference schema) -> dict of device tensors.  Integer columns become int32 with
    missing values (pandas NaN) as NULL_I32; ``rating`` becomes float64 with NaN for missing."""
    out, bounds = {}, {}
    for name in COLUMNS:
        col = df[name].to_numpy()
        if name == "rating":
            out[name] = torch.as_tensor(np.asarray(col, dtype=np.float64), device=device)
            continue
        if col.dtype.kind == "f":
            bad = ~np.isfinite(col)
            if np.any(col[~bad] != np.floor(col[~bad])) or np.any(np.abs(col[~bad]) >= 2 ** 31):
                raise ValueError("column %s holds non-integer values" % name)
            ints = np.where(bad, NULL_I32, np.where(bad, 0, col)).astype(np.int64)
        else:
            ints = col.astype(np.int64)
            if ints.size and (ints.max() >= 2 ** 31 or ints.min() <= NULL_I32):
                raise ValueError("column %s does not fit int32" % name)
        out[name] = torch.as_tensor(ints.astype(np.int32), device=device)
        if name in ("user_id", "anime_id"):      # the id bounds while the column is still on the host
            valid = ints[ints != NULL_I32]
            bounds[name] = max(int(valid.max()) + 1, 1) if valid.size else 1
    return Columns(out, bounds=bounds)


def _bound(t):
    """Exclusive upper bound of the non-missing ids of a column (size of its direct-index table)."""
    return max(int(t.max()) + 1, 1) if t.numel() else 1


def _aligned(t):
    """Contiguous and 16-byte aligned (the kernels read four rows per lane): a sliced view is copied."""
    t = t.contiguous()
    return t if t.data_ptr() % 16 == 0 else t.clone()


def preprocess_columns(cols, num_reviews, drop_unwatched=False, drop_plan=False, drop_half_watched=False):
    """drop_useless (+ drop_half_watched) + scale_ratings on device columns.
    Returns a dict of device tensors with the surviving rows in their original order; its ``bounds`` are the
    input's id bounds (the surviving ids are a subset), so a following ``encode_columns`` need not scan for them."""
    _need_gpu()
    lib = _lib.load()
    u, a = _aligned(cols["user_id"]), _aligned(cols["anime_id"])
    r = _aligned(cols["rating"])
    s, e = _aligned(cols["watching_status"]), _aligned(cols["watched_episodes"])
    assert u.dtype == a.dtype == s.dtype == e.dtype == torch.int32 and r.dtype == torch.float64
    n = int(u.numel())
    dev = u.device
    if n == 0:
        out = Columns({k: cols[k][:0] for k in COLUMNS})
        if drop_half_watched:
            out["max_eps"], out["half_eps"] = a[:0], r[:0]
        return out
    known = getattr(cols, "bounds", {})
    ub, ab = known.get("user_id"), known.get("anime_id")
    if not (ub and ab):          # both maxima in one pass and one readback
        mx = torch.empty(2, dtype=torch.int32, device=dev)
        _lib.check(lib.anirec_ingest_id_max(_lib.ptr(u), _lib.ptr(a), n, _lib.ptr(mx), _stream()), "anirec_ingest_id_max")
        mu, ma = mx.tolist()
        ub, ab = ub or max(mu + 1, 1), ab or max(ma + 1, 1)
    opts = _lib.IngestOpts(int(num_reviews), int(bool(drop_unwatched)), int(bool(drop_plan)),
                           int(bool(drop_half_watched)), int(ub), int(ab))
    ou, oa = torch.empty_like(u), torch.empty_like(a)
    orr, os_, oe = torch.empty_like(r), torch.empty_like(s), torch.empty_like(e)
    res = torch.zeros(2, dtype=torch.int64, device=dev)   # row count | error flag: one readback
    n_out, err = res[0:1], res[1:2].view(torch.int32)[0:1]
    ws = torch.empty(int(lib.anirec_ingest_workspace_bytes(n, opts.user_id_bound, opts.anime_id_bound)),
                     dtype=torch.uint8, device=dev)
    _lib.check(lib.anirec_ingest_preprocess(_lib.ptr(u), _lib.ptr(a), _lib.ptr(r), _lib.ptr(s), _lib.ptr(e), n,
                                            C.byref(opts), _lib.ptr(ou), _lib.ptr(oa), _lib.ptr(orr),
                                            _lib.ptr(os_), _lib.ptr(oe), _lib.ptr(n_out), _lib.ptr(err),
                                            _lib.ptr(ws), ws.numel(), _stream()), "anirec_ingest_preprocess")
    half = None
    if drop_half_watched:        # the reference's frame keeps max_eps / half_eps for this flag (preprocess.py:99-104)
        half = (torch.empty_like(a), torch.empty_like(r))
        _lib.check(lib.anirec_ingest_half_columns(_lib.ptr(oa), _lib.ptr(n_out), n, C.byref(opts), _lib.ptr(half[0]),
                                                  _lib.ptr(half[1]), _lib.ptr(ws), ws.numel(), _stream()),
                   "anirec_ingest_half_columns")
    m, bad, first_nan = torch.cat([n_out, err.to(torch.int64), torch.isnan(orr[:1]).to(torch.int64)]).tolist()
    if bad:
        raise ValueError("negative user_id / anime_id in the rating table")
    if m and first_nan:
        # max == min: the reference's scale_ratings raises here too (preprocess.py:115, Python floats)
        raise ZeroDivisionError("float division by zero (all surviving ratings are equal)")
    out = Columns({"user_id": ou[:m], "anime_id": oa[:m], "rating": orr[:m], "watching_status": os_[:m],
                   "watched_episodes": oe[:m]},
                  bounds={"user_id": opts.user_id_bound, "anime_id": opts.anime_id_bound})
    if half is not None:
        out["max_eps"], out["half_eps"] = half[0][:m], half[1][:m]
    return out


def encode_ids(ids, bound=None):
    """``Series.unique()`` encoding on the GPU: (index int32 [n], uniques int32 [n_unique]).
    ``bound``: an exclusive upper bound of the ids if the caller knows one (default: max + 1, one pass more)."""
    _need_gpu()
    lib = _lib.load()
    ids = _aligned(ids)
    assert ids.dtype == torch.int32
    n = int(ids.numel())
    dev = ids.device
    if n == 0:
        return ids.clone(), ids.clone()
    bound = int(bound) if bound else _bound(ids)
    idx = torch.empty_like(ids)
    uniq = torch.empty(min(n, bound), dtype=torch.int32, device=dev)
    res = torch.zeros(2, dtype=torch.int64, device=dev)   # number of ids | error flag: one readback
    n_u, err = res[0:1], res[1:2].view(torch.int32)[0:1]
    ws = torch.empty(int(lib.anirec_ingest_encode_workspace_bytes(n, bound)), dtype=torch.uint8, device=dev)
    _lib.check(lib.anirec_ingest_encode(_lib.ptr(ids), n, bound, _lib.ptr(idx), _lib.ptr(uniq), _lib.ptr(n_u),
                                        _lib.ptr(err), _lib.ptr(ws), ws.numel(), _stream()),
               "anirec_ingest_encode")
    count, bad = res.tolist()
    if bad & 0xFFFFFFFF:
        raise ValueError("negative id in the column to encode")
    return idx, uniq[:count]


@dataclass
class EncodedRatings:
    """get_df() on the GPU: encoded (and optionally shuffled) rating columns + the id tables."""
    user: torch.Tensor       # int32 [N] dense user index
    anime: torch.Tensor      # int32 [N] dense anime index
    rating: torch.Tensor     # float64 [N]
    user_ids: torch.Tensor   # index -> original user_id
    anime_ids: torch.Tensor  # index -> original anime_id

    # the part of data.RatingTable's interface trainer.fit uses: a table that never leaves HBM trains as it is
    @property
    def n_users(self):
        return int(self.user_ids.numel())

    @property
    def n_anime(self):
        return int(self.anime_ids.numel())

    def __len__(self):
        return int(self.user.numel())

    def split(self, test_size):
        """(train, test) index ranges: the last ``test_size`` shuffled rows are held out (neural_network.py:161-169)."""
        n_train = len(self) - int(test_size)
        if n_train <= 0:
            raise ValueError("test_size %d leaves no training rows out of %d" % (test_size, len(self)))
        return slice(0, n_train), slice(n_train, len(self))


def encode_columns(cols, shuffle=True, random_state=42) -> EncodedRatings:
    """neural_network.py:41-60: encode both id columns by first appearance, then shuffle the rows
    with ``df.sample(frac=1, random_state=42)`` (the permutation itself is NumPy's MT19937 stream,
    generated on the host; the three columns are permuted on the GPU)."""
    from . import data, ops
    known = getattr(cols, "bounds", {})
    ui, user_ids = encode_ids(cols["user_id"], known.get("user_id"))
    ai, anime_ids = encode_ids(cols["anime_id"], known.get("anime_id"))
    r = cols["rating"]
    if shuffle and ui.numel():
        perm = torch.as_tensor(data.shuffle_order(int(ui.numel()), random_state), device=ui.device)
        ui, ai = ui[perm], ai[perm]
        r = r[perm]
    return EncodedRatings(ui, ai, r, user_ids, anime_ids)


def encode_frame(df, shuffle=True, random_state=42, min_ratings=None, device="cuda:0"):
    """``data.encode_frame`` (the reference's get_df) with the id encoding done on the GPU; returns the
    same ``data.RatingTable`` of NumPy columns (int64 indices, float64 ratings), value for value."""
    from . import data
    if min_ratings:
        n_ratings = df["user_id"].value_counts(dropna=True)
        df = df[df["user_id"].isin(n_ratings[n_ratings >= int(min_ratings)].index)]
    ids, dtypes = {}, {}
    for name in ("user_id", "anime_id"):
        col = df[name].to_numpy()
        dtypes[name] = col.dtype
        if col.dtype.kind not in "iu" or (col.size and (col.max() >= 2 ** 31 or col.min() < 0)):
            raise ValueError("%s must hold non-negative integers below 2^31 (the GPU encoder's id tables are "
                             "direct-indexed); got dtype %s" % (name, col.dtype))
        ids[name] = torch.as_tensor(col.astype(np.int32), device=device)
    ui, user_ids = encode_ids(ids["user_id"])
    ai, anime_ids = encode_ids(ids["anime_id"])
    u, a = ui.cpu().numpy().astype(np.int64), ai.cpu().numpy().astype(np.int64)
    r = df["rating"].to_numpy()
    if shuffle:
        order = data.shuffle_order(len(df), random_state)
        u, a, r = u[order], a[order], r[order]
    return data.RatingTable(u, a, r, user_ids.cpu().numpy().astype(dtypes["user_id"]),
                            anime_ids.cpu().numpy().astype(dtypes["anime_id"]))


def load_user_stats(path, **kw):
    import pandas as pd
    return encode_frame(pd.read_parquet(path), **kw)
