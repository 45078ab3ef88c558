"""Parity of the GPU ingest path (anirec_ingest_*) with the pandas restatement of the reference's
preprocess step and id encoding (oracle/ingest_oracle.py).  Bit-exact: integer columns, order,
float64 ratings, indices and the order of the unique-id tables."""
import numpy as np
import pandas as pd
import pytest
import torch

from oracle import ingest_oracle as orc

pytestmark = pytest.mark.gpu


def _raw_frame(n, n_users, n_anime, seed, nulls=True, dups=True, grouped=False):
    rng = np.random.default_rng(seed)
    users = np.sort(rng.choice(np.arange(1, 5 * n_users), n_users, replace=False))
    animes = rng.choice(np.arange(1, 3 * n_anime), n_anime, replace=False)
    # skewed activity: some users fall below any sensible num_reviews
    w = rng.lognormal(0, 1.2, n_users)
    df = pd.DataFrame({
        "user_id": users[rng.choice(n_users, n, p=w / w.sum())],
        "anime_id": animes[rng.integers(0, n_anime, n)],
        "rating": rng.integers(0, 11, n),
        "watching_status": rng.choice([1, 2, 3, 4, 6], n),
        "watched_episodes": rng.integers(0, 30, n),
    })
    if dups:  # exact duplicate rows scattered through the frame (first occurrence must win)
        src = rng.integers(0, n, n // 20)
        dst = rng.integers(0, n, n // 20)
        df.iloc[dst] = df.iloc[src].to_numpy()
    if nulls:
        df = df.astype({"user_id": "float64", "anime_id": "float64", "rating": "float64",
                        "watching_status": "float64", "watched_episodes": "float64"})
        for col in df.columns:
            df.loc[rng.integers(0, n, max(1, n // 500)), col] = np.nan
    if grouped:  # the raw animelist: one block of rows per user (duplicates stay inside their user's block)
        df = df.sort_values("user_id", kind="stable").reset_index(drop=True)
    return df


def _gpu_preprocess(df, **kw):
    from anime_recommendations_amd import ingest
    cols = ingest.frame_to_columns(df)
    return {k: v.cpu().numpy() for k, v in ingest.preprocess_columns(cols, **kw).items()}


def _check(df, num_reviews, **flags):
    want = orc.preprocess(df, num_reviews, flags.get("drop_unwatched", False), flags.get("drop_plan", False),
                          flags.get("drop_half_watched", False))
    got = _gpu_preprocess(df, num_reviews=num_reviews, **flags)
    assert len(got["user_id"]) == len(want)
    for col in ("user_id", "anime_id", "watching_status", "watched_episodes"):
        np.testing.assert_array_equal(got[col].astype(np.int64), want[col].to_numpy().astype(np.int64), err_msg=col)
    w = want["rating"].to_numpy().astype(np.float64)
    assert got["rating"].dtype == np.float64
    np.testing.assert_array_equal(got["rating"].view(np.uint64), w.view(np.uint64))
    return want


@pytest.mark.parametrize("flags", [dict(), dict(drop_unwatched=True), dict(drop_plan=True),
                                   dict(drop_unwatched=True, drop_plan=True, drop_half_watched=True),
                                   dict(drop_half_watched=True)])
def test_preprocess_matches_pandas(flags):
    df = _raw_frame(200_000, 900, 700, seed=3)
    want = _check(df, 150, **flags)
    assert 0 < len(want) < len(df)


@pytest.mark.parametrize("flags", [dict(), dict(drop_plan=True), dict(drop_unwatched=True, drop_half_watched=True)])
@pytest.mark.parametrize("shape", [(200_000, 900), (200_000, 7), (70_000, 1), (16_384 * 3, 48), (16_385, 300)])
def test_preprocess_grouped_by_user(flags, shape):
    """The layout of the raw animelist: users in blocks.  Blocks inside one 16 384-row chunk take the LDS path
    of k_ing_front, blocks that straddle a chunk boundary (or are longer than a chunk: 7 users, 1 user) the
    list path; both must give pandas' answer, including duplicates whose first occurrence sits in another chunk."""
    n, n_users = shape
    df = _raw_frame(n, n_users, 700, seed=21 + n_users, grouped=True)
    want = _check(df, max(2, n // n_users // 2), **flags)
    assert len(want) > 0
    # and with a threshold no straddling user passes
    _check(df, n // n_users * 2, **flags) if n_users > 1 else None


def test_preprocess_grouped_many_duplicates():
    """Every row appears 3 times inside its user's block (first occurrence wins, order kept)."""
    df = _raw_frame(40_000, 150, 90, seed=33, nulls=False, dups=False, grouped=True)
    df = pd.concat([df, df, df], ignore_index=True).sort_values("user_id", kind="stable").reset_index(drop=True)
    want = _check(df, 50)
    assert len(want) <= 40_000


def test_preprocess_no_nulls_integer_columns():
    df = _raw_frame(50_000, 300, 400, seed=5, nulls=False)
    _check(df, 100)


def test_preprocess_edge_cases():
    # everything dropped
    df = _raw_frame(5_000, 200, 100, seed=7)
    got = _gpu_preprocess(df, num_reviews=10 ** 6)
    assert all(len(v) == 0 for v in got.values())
    # single row / constant rating: max == min, the reference's scale_ratings raises ZeroDivisionError
    one = pd.DataFrame({"user_id": [7], "anime_id": [3], "rating": [5.0], "watching_status": [2], "watched_episodes": [1]})
    with pytest.raises(ZeroDivisionError):
        orc.preprocess(one, 1)
    with pytest.raises(ZeroDivisionError):
        _gpu_preprocess(one, num_reviews=1)
    # the duplicate of row 0 sits at the very end; a tile-boundary sized frame
    df = _raw_frame(4096 * 3, 50, 60, seed=9, nulls=False, dups=False)
    df = pd.concat([df, df.iloc[[0]]], ignore_index=True)
    _check(df, 1)
    df = _raw_frame(4096 * 2, 50, 60, seed=10, nulls=False)
    _check(df, 20, drop_half_watched=True)


def test_encode_matches_series_unique():
    from anime_recommendations_amd import ingest
    rng = np.random.default_rng(11)
    for n, hi in ((1, 10), (4096, 50), (300_000, 70_000), (300_001, 5)):
        ids = rng.integers(0, hi, n).astype(np.int32) * 3 + 1
        idx, uniq = ingest.encode_ids(torch.as_tensor(ids, device="cuda"))
        want_idx, want_uniq = orc.encode(pd.Series(ids))
        np.testing.assert_array_equal(idx.cpu().numpy(), want_idx)
        np.testing.assert_array_equal(uniq.cpu().numpy(), want_uniq)
        # a caller-supplied bound (any upper bound of the ids: preprocess hands its input's on) changes nothing
        idx2, uniq2 = ingest.encode_ids(torch.as_tensor(ids, device="cuda"), bound=3 * hi + 977)
        assert torch.equal(idx2, idx) and torch.equal(uniq2, uniq)


def test_preprocess_then_encode_feeds_training_layout():
    """End to end against data.encode_frame (the host path the trainer used so far)."""
    from anime_recommendations_amd import data, ingest
    df = _raw_frame(120_000, 500, 600, seed=13)
    want_df = orc.preprocess(df, 100)
    cols = ingest.preprocess_columns(ingest.frame_to_columns(df), num_reviews=100)
    enc = ingest.encode_columns(cols, shuffle=True, random_state=42)
    ref = data.encode_frame(want_df.astype({"user_id": "int64", "anime_id": "int64"}), shuffle=True, random_state=42)
    np.testing.assert_array_equal(enc.user.cpu().numpy(), ref.user)
    np.testing.assert_array_equal(enc.anime.cpu().numpy(), ref.anime)
    np.testing.assert_array_equal(enc.rating.cpu().numpy(), ref.rating)
    np.testing.assert_array_equal(enc.user_ids.cpu().numpy(), ref.user_ids)
    np.testing.assert_array_equal(enc.anime_ids.cpu().numpy(), ref.anime_ids)


@pytest.mark.parametrize("grouped", [True, False])
def test_fullsize_properties_109m_rows(grouped):
    """BASELINE size (109 M rows): no oracle run, size-independent properties instead —
    idempotence, order preservation, per-user counts >= num_reviews, no duplicate rows.  Grouped by user (the raw
    animelist: the LDS path) and in random order (every row through the list and the global table)."""
    from anime_recommendations_amd import ingest
    n, n_users, n_anime = 109_000_000, 350_000, 18_000
    g = torch.Generator(device="cuda")
    g.manual_seed(17)
    users = torch.randint(0, n_users, (n,), generator=g, device="cuda", dtype=torch.int32)
    cols = {
        "user_id": torch.sort(users)[0] if grouped else users,
        "anime_id": torch.randint(0, n_anime, (n,), generator=g, device="cuda", dtype=torch.int32),
        "rating": torch.randint(0, 11, (n,), generator=g, device="cuda", dtype=torch.int32).double(),
        "watching_status": torch.randint(1, 7, (n,), generator=g, device="cuda", dtype=torch.int32),
        "watched_episodes": torch.randint(0, 3, (n,), generator=g, device="cuda", dtype=torch.int32),
    }
    out = ingest.preprocess_columns(cols, num_reviews=300, drop_plan=True)
    m = int(out["user_id"].numel())
    assert 0 < m < n
    assert int((out["watching_status"] == 6).sum()) == 0
    cnt = torch.bincount(out["user_id"].long(), minlength=n_users)
    assert int(cnt[cnt > 0].min()) >= 300
    assert float(out["rating"].min()) == 0.0 and float(out["rating"].max()) == 1.0
    # no duplicate rows survive: (user, anime, rating*10, status, episodes) packed into one int64 key
    key = ((out["user_id"].long() * n_anime + out["anime_id"].long()) * 11 +
           (out["rating"] * 10).round().long()) * 8 * 4 + out["watching_status"].long() * 4 + out["watched_episodes"].long()
    assert int(torch.unique(key).numel()) == m
    # idempotent (second scaling is (x - 0) / (1 - 0))
    again = ingest.preprocess_columns(out, num_reviews=300, drop_plan=True)
    for k in out:
        assert torch.equal(again[k], out[k]), k
    # encoding: indices are a bijection onto first-appearance order
    idx, uniq = ingest.encode_ids(out["user_id"])
    assert torch.equal(uniq[idx.long()], out["user_id"])
    first = torch.full((int(uniq.numel()),), m, dtype=torch.int64, device="cuda")
    first.scatter_reduce_(0, idx.long(), torch.arange(m, device="cuda"), reduce="amin")
    assert bool((first[1:] > first[:-1]).all())


def test_encode_frame_gpu_equals_host_encode_frame():
    """The RatingTable the neural_network component trains on: GPU id encoding == data.encode_frame."""
    from anime_recommendations_amd import data, ingest
    df = data.synth_user_stats(n_users=400, n_anime=300, n_ratings=30_000, seed=5)
    for kw in (dict(), dict(shuffle=False), dict(min_ratings=60)):
        ref = data.encode_frame(df, **kw)
        got = ingest.encode_frame(df, **kw)
        for f in ("user", "anime", "rating", "user_ids", "anime_ids"):
            a, b = getattr(got, f), getattr(ref, f)
            assert a.dtype == b.dtype and np.array_equal(a, b), f


def test_unaligned_views_are_accepted_by_the_wrappers_and_rejected_by_the_c_entry():
    """The kernels read 16 bytes at a time: the C entry points return ANIREC_EINVAL for a misaligned column, the
    Python wrappers copy a sliced view first."""
    import ctypes as C
    from anime_recommendations_amd import _lib, ingest
    df = _raw_frame(20_001, 90, 70, seed=3, nulls=False)
    cols = ingest.frame_to_columns(df)
    view = {k: v[1:] for k, v in cols.items()}               # 4 / 8 bytes past an aligned address
    assert any(v.data_ptr() % 16 for v in view.values())
    got = ingest.preprocess_columns(view, num_reviews=20)
    want = orc.preprocess(df.iloc[1:].reset_index(drop=True), 20)
    np.testing.assert_array_equal(got["user_id"].cpu().numpy(), want["user_id"].to_numpy())
    idx, uniq = ingest.encode_ids(cols["user_id"][1:])
    want_idx, want_uniq = orc.encode(pd.Series(cols["user_id"][1:].cpu().numpy()))
    np.testing.assert_array_equal(idx.cpu().numpy(), want_idx)
    lib = _lib.load()
    ids = cols["user_id"]
    n = ids.numel() - 1
    ws = torch.empty(int(lib.anirec_ingest_encode_workspace_bytes(n, 1000)), dtype=torch.uint8, device="cuda")
    out = torch.empty(n + 4, dtype=torch.int32, device="cuda")
    z = torch.zeros(4, dtype=torch.int64, device="cuda")
    rc = lib.anirec_ingest_encode(C.c_void_p(ids.data_ptr() + 4), n, 1000, _lib.ptr(out), _lib.ptr(out), _lib.ptr(z),
                                  _lib.ptr(z[2:]), _lib.ptr(ws), ws.numel(), None)
    assert rc == _lib.EINVAL if hasattr(_lib, "EINVAL") else rc < 0


def test_preprocess_one_row_repeated_thousands_of_times():
    """A chunk whose 8 192 rows are all the same row (every insert lands on one LDS slot), a run of them crossing
    chunk boundaries, and the same again through the list path (the repeated user also appears at the far end)."""
    base = _raw_frame(3_000, 40, 50, seed=4, nulls=False, dups=False, grouped=True)
    row = base.iloc[[1234]]
    rep = pd.concat([row] * 20_000, ignore_index=True)
    for tail in (False, True):
        parts = [base.iloc[:1234], rep, base.iloc[1234:]] + ([row] if tail else [])
        df = pd.concat(parts, ignore_index=True)
        want = _check(df, 10)
        assert len(want) <= len(base)


def _blocks_frame(lengths, seed, n_anime=400, dups=0.02):
    """One block of rows per user with the given lengths, in order; duplicates inside their user's block."""
    rng = np.random.default_rng(seed)
    users = np.repeat(np.arange(len(lengths)) * 7 + 3, lengths)
    n = len(users)
    df = pd.DataFrame({"user_id": users, "anime_id": rng.integers(1, n_anime, n), "rating": rng.integers(0, 11, n),
                       "watching_status": rng.choice([1, 2, 3, 4, 6], n), "watched_episodes": rng.integers(0, 30, n)})
    start = np.repeat(np.cumsum(lengths) - lengths, lengths)
    src = rng.integers(0, n, int(n * dups))
    dst = np.minimum(src + rng.integers(1, 200, len(src)), start[src] + np.repeat(lengths, lengths)[src] - 1)
    df.iloc[dst] = df.iloc[src].to_numpy()          # still the same user: dst stays inside src's block
    return df


@pytest.mark.parametrize("reach", [0, 1, 4095, 4096, 4097, 9000])
def test_preprocess_user_blocks_around_a_chunk_boundary(reach):
    """k_ing_front: the workgroup of the chunk a user starts in reads on to the user's last row, at most 4 096 rows
    past its 8 192-row chunk.  A block that ends exactly at the boundary, one row / 4 095 rows past it (owned),
    4 096 and more (nobody's: the list), and a block over three chunks — with duplicates whose first occurrence lies on
    the other side of the boundary, and thresholds on both sides of every block's count."""
    lengths = [5000, 3192 - 700, 700 + reach, 300, 8192 * 2 + 11, 250, 6000]
    assert sum(lengths[:2]) + 700 == 8192           # the third block starts 700 rows before the first boundary
    df = _blocks_frame(lengths, seed=50 + reach)
    for num_reviews in (1, 280, 650, 5100):
        want = _check(df, num_reviews, drop_plan=True)
    assert len(want) > 0


def test_preprocess_rows_of_another_user_inside_a_block():
    """Almost grouped: single rows of user A inside the blocks of other users, near and across chunk boundaries (the
    bench table's duplicates do this).  A is then owned only if its first and last row allow it; whatever the
    workgroups decide, pandas' answer must come out."""
    lengths = [3000, 5000, 400, 7900, 300, 8100, 2000]
    df = _blocks_frame(lengths, seed=77)
    rng = np.random.default_rng(78)
    n = len(df)
    for pos in (8190, 8191, 8192, 8193, 8192 + 390, 16383, 16384, 16385, n - 1, 10, 12000):
        src = int(rng.integers(0, n))
        df.iloc[pos] = df.iloc[src].to_numpy()     # user, anime, ... of a row far away
    for num_reviews in (1, 350, 4000):
        _check(df, num_reviews)
    _check(df, 300, drop_unwatched=True, drop_half_watched=True)


def test_encode_id_space_at_the_lds_table_limit():
    """k_enc_*_lds holds the id tables of up to 32 768 ids in LDS; one id more takes the global-table kernels."""
    from anime_recommendations_amd import ingest
    rng = np.random.default_rng(5)
    for bound in (32767, 32768, 32769):
        ids = rng.integers(0, bound, 200_003).astype(np.int32)
        ids[-1] = bound - 1
        idx, uniq = ingest.encode_ids(torch.as_tensor(ids, device="cuda"))
        want_idx, want_uniq = orc.encode(pd.Series(ids))
        np.testing.assert_array_equal(idx.cpu().numpy(), want_idx)
        np.testing.assert_array_equal(uniq.cpu().numpy(), want_uniq)


@pytest.mark.parametrize("seed", range(10))
def test_preprocess_random_layouts_against_pandas(seed):
    """Differential fuzz of the ownership logic of k_ing_front: tables built from random pieces — user blocks of a few
    rows up to well over a chunk + its reach (12 000 rows), stretches in random order, the same user in several
    places, single rows of other users dropped into blocks, rows with missing values, duplicates near and far —
    with random flags and thresholds.  Whatever the workgroups decide about who owns which user, pandas' answer must
    come out, bit for bit."""
    rng = np.random.default_rng(1000 + seed)
    n_anime = int(rng.integers(30, 3000))
    parts, uid = [], 1
    target = int(rng.integers(20_000, 70_000))
    total = 0
    while total < target:
        kind = rng.random()
        if kind < 0.7:                                   # one user's block
            ln = int(min(12_000, max(1, rng.lognormal(5.5, 1.6))))
            users = np.full(ln, uid)
            uid += int(rng.integers(1, 4))
        elif kind < 0.85:                                # a stretch in random order over a few (partly earlier) users
            ln = int(rng.integers(50, 4000))
            pool = rng.integers(max(1, uid - 6), uid + 4, int(rng.integers(2, 9)))
            users = rng.choice(pool, ln)
            uid += 4
        else:                                            # an earlier user comes back
            ln = int(rng.integers(1, 600))
            users = np.full(ln, int(rng.integers(1, uid + 1)))
        parts.append(users)
        total += ln
    users = np.concatenate(parts)
    n = len(users)
    df = pd.DataFrame({"user_id": users, "anime_id": rng.integers(1, n_anime, n), "rating": rng.integers(0, 11, n),
                       "watching_status": rng.choice([1, 2, 3, 4, 6], n), "watched_episodes": rng.integers(0, 30, n)})
    for _ in range(int(rng.integers(0, 40))):            # single rows copied elsewhere (user id and all)
        df.iloc[int(rng.integers(0, n))] = df.iloc[int(rng.integers(0, n))].to_numpy()
    src = rng.integers(0, n, n // 30)                    # duplicates a few rows further on
    dst = np.minimum(src + rng.integers(1, 80, len(src)), n - 1)
    df.iloc[dst] = df.iloc[src].to_numpy()
    if rng.random() < 0.6:
        df = df.astype("float64")
        for col in df.columns:
            df.loc[rng.integers(0, n, int(rng.integers(1, 30))), col] = np.nan
    if rng.random() < 0.3:                               # non-integer ratings: the rating-key fast path must not bite
        df["rating"] = df["rating"].astype("float64") * 0.37
    flags = dict(drop_unwatched=bool(rng.random() < 0.4), drop_plan=bool(rng.random() < 0.5),
                 drop_half_watched=bool(rng.random() < 0.3))
    counts = df["user_id"].value_counts()
    for num_reviews in (1, int(counts.median()) + 1, int(counts.quantile(0.9)) + 1):
        try:
            want = orc.preprocess(df, num_reviews, flags["drop_unwatched"], flags["drop_plan"], flags["drop_half_watched"])
        except ZeroDivisionError:
            with pytest.raises(ZeroDivisionError):
                _gpu_preprocess(df, num_reviews=num_reviews, **flags)
            continue
        if len(want) == 0:
            got = _gpu_preprocess(df, num_reviews=num_reviews, **flags)
            assert len(got["user_id"]) == 0
            continue
        _check(df, num_reviews, **flags)


@pytest.mark.parametrize("seed", range(6))
def test_encode_random_columns_against_series_unique(seed):
    """Differential fuzz of the id encoders: columns with long runs (a table grouped by the column: unread atomics at
    the run ends), without runs (read before update), mixtures, id spaces on both sides of the 32 768-id LDS-table
    limit, lengths that are multiples of nothing, caller-supplied loose bounds — always Series.unique() order."""
    from anime_recommendations_amd import ingest
    rng = np.random.default_rng(500 + seed)
    for _ in range(4):
        n = int(rng.integers(1, 400_000))
        hi = int(rng.choice([3, 200, 18_000, 32_768, 32_769, 90_000, 400_000]))
        kind = rng.random()
        if kind < 0.35:
            ids = rng.integers(0, hi, n)
        elif kind < 0.7:                                  # runs: sorted, or blocks in random order of ids
            ids = np.sort(rng.integers(0, hi, n)) if rng.random() < 0.5 else np.repeat(
                rng.permutation(hi)[:max(1, n // 50)], 50)[:n]
            if len(ids) < n:
                ids = np.concatenate([ids, rng.integers(0, hi, n - len(ids))])
        else:                                             # runs with strangers sprinkled in
            ids = np.sort(rng.integers(0, hi, n))
            at = rng.integers(0, n, max(1, n // 100))
            ids[at] = rng.integers(0, hi, len(at))
        ids = ids.astype(np.int32)
        want_idx, want_uniq = orc.encode(pd.Series(ids))
        t = torch.as_tensor(ids, device="cuda")
        for bound in (None, int(ids.max()) + 1 + int(rng.integers(0, 5000))):
            idx, uniq = ingest.encode_ids(t, bound=bound)
            np.testing.assert_array_equal(idx.cpu().numpy(), want_idx)
            np.testing.assert_array_equal(uniq.cpu().numpy(), want_uniq)
