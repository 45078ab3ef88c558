"""GPU parity tests of the inference hot path (row-normalise, cosine top-k, predict)."""
import numpy as np
import pytest
import torch

from oracle import anirec_oracle as orc

pytestmark = pytest.mark.gpu


def test_rownorm_matches_numpy_within_2ulp_and_nan_on_zero_row():
    from anime_recommendations_amd import ops
    W = np.random.default_rng(0).normal(0, 0.05, (5000, 128)).astype(np.float32)
    W[17] = 0
    with np.errstate(invalid="ignore", divide="ignore"):
        ref = orc.rownorm(W)
    got = ops.rownorm(torch.from_numpy(W)).cpu().numpy()
    assert np.isnan(got[17]).all()
    m = ~np.isnan(ref)
    np.testing.assert_allclose(got[m], ref[m], rtol=3e-7, atol=0)


def test_cosine_scores_are_the_defined_fma_chain_bitwise():
    from anime_recommendations_amd import ops
    W = np.random.default_rng(1).normal(0, 0.05, (777, 128)).astype(np.float32)
    Wh = ops.rownorm(torch.from_numpy(W))
    s = ops.cosine_scores(Wh, 5).cpu().numpy()
    ref = orc.dot_chain_f32(Wh.cpu().numpy(), Wh.cpu().numpy()[5])
    assert (s == ref).mean() > 0.999           # numpy emulation can double-round on rare ties
    np.testing.assert_allclose(s, ref, atol=1.2e-7)


@pytest.mark.parametrize("n,k", [(1000, 10), (17560, 100), (333, 128)])
def test_cosine_topk_indices_exact(n, k):
    from anime_recommendations_amd import ops
    rng = np.random.default_rng(2)
    W = rng.normal(0, 0.05, (n, 128)).astype(np.float32)
    W[11] = W[4]                               # exact duplicates -> exact score ties
    W[12] = W[4]
    Wh = ops.rownorm(torch.from_numpy(W))
    queries = [4, 0, n - 1, 11] + list(rng.integers(0, n, 12))
    idx, sim = ops.cosine_topk(Wh, queries, k)
    # oracle on the GPU's own fixed-order scores: selection + tie rule must be exact
    Whn = Wh.cpu().numpy()
    idx, sim = idx.cpu().numpy(), sim.cpu().numpy()
    for j, q in enumerate(queries):
        s = ops.cosine_scores(Wh, q).cpu().numpy()
        oi, os_ = orc.topk_desc(s, k, exclude=q)
        assert (idx[j, :len(oi)] == oi).all(), (j, q)
        assert (sim[j, :len(oi)] == os_).all()
        # and against an independent fp64 ranking wherever the gaps exceed fp32 noise
        s64 = Whn.astype(np.float64) @ Whn[q].astype(np.float64)
        s64[q] = -np.inf
        o64 = np.argsort(-s64, kind="stable")[:k]
        gaps = np.abs(np.diff(s64[np.argsort(-s64, kind="stable")[:k + 1]]))
        if gaps.min() > 1e-6:
            assert (idx[j] == o64).all()
    assert 11 in idx[0][:2] and 12 in idx[0][:2] and idx[0][0] == 11   # ties -> ascending index


def test_cosine_topk_mask_short_rows_and_nan():
    from anime_recommendations_amd import ops
    rng = np.random.default_rng(3)
    n = 500
    W = rng.normal(0, 0.05, (n, 128)).astype(np.float32)
    W[9] = 0                                    # NaN row
    Wh = ops.rownorm(torch.from_numpy(W))
    keep = np.zeros(n, np.uint8)
    keep[:7] = 1
    keep[9] = 1
    idx, sim = ops.cosine_topk(Wh, [2, 300], 10, keep=keep)
    idx, sim = idx.cpu().numpy(), sim.cpu().numpy()
    for j, q in enumerate([2, 300]):
        s = ops.cosine_scores(Wh, q).cpu().numpy()
        oi, _ = orc.topk_desc(s, 10, exclude=q, mask=keep.astype(bool))
        assert (idx[j, :len(oi)] == oi).all()
        assert (idx[j, len(oi):] == -1).all()
        assert oi[-1] == 9                      # NaN candidate ranks last
    # no exclusion: the query is its own best match
    idx2, _ = ops.cosine_topk(Wh, [2], 3, exclude_self=False)
    assert idx2.cpu().numpy()[0, 0] == 2


def test_predict_pairs_grid_topk_match_oracle():
    from anime_recommendations_amd import ops
    rng = np.random.default_rng(4)
    n_u, n_a = 700, 1234
    U = rng.normal(0, 0.05, (n_u, 128)).astype(np.float32)
    A = rng.normal(0, 0.05, (n_a, 128)).astype(np.float32)
    head = dict(w=1.3, b=0.1, gamma=0.9, beta=-0.2, mov_mean=0.05, mov_var=0.4)
    oh = orc.new_head(**head)
    tU, tA = torch.from_numpy(U).cuda(), torch.from_numpy(A).cuda()
    ui = rng.integers(0, n_u, 3000)
    ai = rng.integers(0, n_a, 3000)
    p = ops.predict_pairs(tU, tA, head, ui, ai).cpu().numpy()
    np.testing.assert_allclose(p, orc.predict_pairs(U, A, oh, ui, ai), atol=1e-5)   # BASELINE bar
    users = [5, 0, 699, 123, 77]
    G = ops.predict_grid(tU, tA, head, users).cpu().numpy()
    Go = orc.predict_grid(U, A, oh, users)
    np.testing.assert_allclose(G, Go, atol=1e-5)
    # top-k with a watched mask
    watched = rng.random((len(users), n_a)) < 0.3
    bits = np.zeros((len(users), (n_a + 31) // 32), np.uint32)
    for j in range(len(users)):
        for a in np.nonzero(watched[j])[0]:
            bits[j, a >> 5] |= np.uint32(1) << np.uint32(a & 31)
    ti, tp = ops.predict_topk(tU, tA, head, users, 10, bits.view(np.int32))
    ti, tp = ti.cpu().numpy(), tp.cpu().numpy()
    for j in range(len(users)):
        oi, op = orc.topk_desc(G[j], 10, mask=~watched[j])
        assert (ti[j] == oi).all()
        assert (tp[j] == op).all()


@pytest.mark.parametrize("n,nq,k", [(1000, 1000, 10), (5000, 700, 100), (17560, 2048, 100), (333, 333, 127)])
def test_mfma_topk_equals_exact_path_bitwise(n, nq, k):
    """f16-MFMA candidates + exact fp32 re-rank must reproduce the exact kernels' lists exactly."""
    from anime_recommendations_amd import ops
    rng = np.random.default_rng(10)
    W = rng.normal(0, 0.05, (n, 128)).astype(np.float32)
    W[11] = W[4]                               # exact score ties
    W[12] = W[4]
    W[20] = 0                                  # NaN row after normalisation
    Wh = ops.rownorm(torch.from_numpy(W))
    queries = rng.permutation(n)[:nq].astype(np.int32)
    queries[:3] = [4, 11, 20]
    ei, es = ops.cosine_topk(Wh, queries, k)
    mi, ms, nfb = ops.cosine_topk_mfma(Wh, queries, k)
    ei, es, mi, ms = (x.cpu().numpy() for x in (ei, es, mi, ms))
    assert (mi == ei).all()
    assert (ms == es)[~np.isnan(es)].all() and (np.isnan(ms) == np.isnan(es)).all()
    assert nfb <= max(2, nq // 50)             # the NaN query row must fall back; hardly anything else


def test_mfma_topk_masks_and_no_self_exclusion():
    from anime_recommendations_amd import ops
    rng = np.random.default_rng(11)
    n = 4000
    Wh = ops.rownorm(torch.from_numpy(rng.normal(0, 0.05, (n, 128)).astype(np.float32)))
    keep = (rng.random(n) < 0.5).astype(np.uint8)
    q = np.arange(0, n, 7, dtype=np.int32)
    for excl in (True, False):
        ei, es = ops.cosine_topk(Wh, q, 20, exclude_self=excl, keep=keep)
        mi, ms, _ = ops.cosine_topk_mfma(Wh, q, 20, exclude_self=excl, keep=keep)
        assert (mi.cpu().numpy() == ei.cpu().numpy()).all() and (ms.cpu().numpy() == es.cpu().numpy()).all()


def test_mfma_topk_dense_cluster_falls_back_but_stays_exact():
    """Hundreds of near-duplicate rows put >256 keys inside the 2-eps window: the kernel must
    flag those queries (never return a wrong list) and the wrapper re-runs them exactly."""
    from anime_recommendations_amd import ops
    rng = np.random.default_rng(12)
    n = 3000
    W = rng.normal(0, 0.05, (n, 128)).astype(np.float32)
    base = W[0].copy()
    W[:600] = base + rng.normal(0, 1e-4, (600, 128)).astype(np.float32)
    Wh = ops.rownorm(torch.from_numpy(W))
    q = np.arange(0, 1200, 3, dtype=np.int32)
    ei, es = ops.cosine_topk(Wh, q, 50)
    mi, ms, nfb = ops.cosine_topk_mfma(Wh, q, 50)
    assert nfb >= 100
    assert (mi.cpu().numpy() == ei.cpu().numpy()).all() and (ms.cpu().numpy() == es.cpu().numpy()).all()


def test_predict_grid_mfma_within_1e5_of_oracle_and_fp32_path():
    from anime_recommendations_amd import ops
    rng = np.random.default_rng(13)
    n_u, n_a = 900, 1777
    U = rng.normal(0, 0.05, (n_u, 128)).astype(np.float32)
    A = rng.normal(0, 0.05, (n_a, 128)).astype(np.float32)
    A[5] *= 40.0                               # wide dynamic range of row norms
    U[7] = 0.0                                 # zero row: l2_normalize clamps, c = 0
    head = dict(w=1.3, b=0.1, gamma=0.9, beta=-0.2, mov_mean=0.05, mov_var=0.4)
    tU, tA = torch.from_numpy(U).cuda(), torch.from_numpy(A).cuda()
    users = np.concatenate([[7, 0, n_u - 1], rng.integers(0, n_u, 300)])
    G = ops.predict_grid_mfma(tU, tA, head, users).cpu().numpy()
    Gf = ops.predict_grid(tU, tA, head, users).cpu().numpy()
    np.testing.assert_allclose(G, Gf, atol=3e-6)
    Go = orc.predict_grid(U, A, orc.new_head(**head), users[:40])
    np.testing.assert_allclose(G[:40], Go, atol=1e-5)      # BASELINE bar


@pytest.mark.parametrize("shape", [(303, 1776), (700, 17984), (1000, 64), (257, 128), (513, 4160), (40, 18000)])
def test_predict_grid_mfma_row_quad_kernel_on_odd_shapes(shape, monkeypatch):
    """n_anime % 4 == 0 takes k_predict_mfma2 (LDS-DMA tiles, 4 rows x 256 B per store; the test above has an n_anime
    that is not: the dword-store kernel).  User counts that end inside a workgroup / a wave, tile counts that do not
    divide into the anime parts, one to seven parts — within 3e-6 of the fp32 kernel, bit-identical whatever the part
    count, and no byte written behind the grid."""
    from anime_recommendations_amd import ops
    n_users, n_a = shape
    rng = np.random.default_rng(n_users + n_a)
    U = torch.from_numpy(rng.normal(0, 0.05, (n_users + 5, 128)).astype(np.float32)).cuda()
    A = torch.from_numpy(rng.normal(0, 0.05, (n_a, 128)).astype(np.float32)).cuda()
    head = dict(w=1.3, b=0.1, gamma=0.9, beta=-0.2, mov_mean=0.05, mov_var=0.4)
    users = rng.integers(0, n_users + 5, n_users).astype(np.int32)
    Gf = ops.predict_grid(U, A, head, users).cpu().numpy()
    first = None
    for parts in (None, "1", "3", "7"):
        if parts is None:
            monkeypatch.delenv("ANIREC_PREDICT_PARTS", raising=False)
        else:
            monkeypatch.setenv("ANIREC_PREDICT_PARTS", parts)
        out = torch.full((n_users + 3, n_a), -7.0, device="cuda")        # rows behind the grid must stay untouched
        ops.predict_grid_mfma(U, A, head, users, out=out[:n_users])
        assert bool((out[n_users:] == -7.0).all())
        np.testing.assert_allclose(out[:n_users].cpu().numpy(), Gf, atol=3e-6)
        first = out[:n_users].clone() if first is None else first
        assert torch.equal(out[:n_users], first), (shape, parts)


def _watched_bits(rng, nq, n_a, frac):
    w = rng.random((nq, n_a)) < frac
    bits = np.zeros((nq, (n_a + 31) // 32), np.uint32)
    for a in range(n_a):
        bits[:, a >> 5] |= (w[:, a].astype(np.uint32) << np.uint32(a & 31))
    return w, bits.view(np.int32)


@pytest.mark.parametrize("head_kw,expect_fallback", [
    (dict(w=1.3, b=0.1, gamma=0.9, beta=-0.2, mov_mean=0.05, mov_var=0.4), False),
    (dict(w=-2.0, b=0.3, gamma=1.1, beta=0.1, mov_mean=-0.02, mov_var=0.9), False),   # negative slope: ranks reversed
    (dict(w=400.0, b=0.0, gamma=1.0, beta=30.0, mov_mean=0.0, mov_var=1.0), True),     # saturated sigmoid: ties -> fallback
    (dict(w=0.0, b=0.3, gamma=1.0, beta=0.1, mov_mean=0.0, mov_var=1.0), True),        # zero slope: every rating equal
])
def test_predict_topk_mfma_equals_exact_path_bitwise(head_kw, expect_fallback):
    """model_recs batched: MFMA candidates + watched mask + exact re-rank == exact kernels, bit for bit."""
    from anime_recommendations_amd import ops
    rng = np.random.default_rng(31)
    n_u, n_a, nq, k = 900, 3000, 600, 10
    U = torch.from_numpy(rng.normal(0, 0.05, (n_u, 128)).astype(np.float32)).cuda()
    A = rng.normal(0, 0.05, (n_a, 128)).astype(np.float32)
    A[7] = A[3]                                   # exact rating ties between anime
    A[9] = 0                                      # zero row: tf l2_normalize gives 0, cosine 0
    A = torch.from_numpy(A).cuda()
    users = rng.permutation(n_u)[:nq].astype(np.int32)
    w, bits = _watched_bits(rng, nq, n_a, 0.3)
    bits_few = bits.copy()
    bits_few[5] = -1                              # user 5 has watched everything: fewer than k candidates
    bits_few[6, :-1] = -1                         # user 6 has 24 unwatched anime at most
    for wb in (None, bits, bits_few):
        ei, ep = ops.predict_topk(U, A, head_kw, users, k, wb)
        mi, mp, nfb = ops.predict_topk_mfma(U, A, head_kw, users, k, wb)
        ei, ep, mi, mp = (x.cpu().numpy() for x in (ei, ep, mi, mp))
        assert (mi == ei).all()
        assert (mp == ep)[~np.isnan(ep)].all() and (np.isnan(mp) == np.isnan(ep)).all()
        if expect_fallback:
            assert nfb > nq // 2
        else:
            assert nfb <= (4 if wb is bits_few else 2)
    # without the fallback the flagged rows are -1 / NaN, never wrong
    mi2, mp2, _ = ops.predict_topk_mfma(U, A, head_kw, users, k, bits, fallback=False)
    mi2 = mi2.cpu().numpy()
    ei, _ = ops.predict_topk(U, A, head_kw, users, k, bits)
    ok = mi2[:, 0] >= 0
    assert (mi2[ok] == ei.cpu().numpy()[ok]).all()


def test_cosine_topk_few_queries_sliced_select_keeps_the_tie_rule():
    """Few queries against many keys (the reference's literal call is ONE query): the keys are cut into
    slices selected by separate workgroups and merged — duplicates straddling slice boundaries, a key
    mask, self exclusion and k = 128 must come out exactly as the single-workgroup rule says."""
    from anime_recommendations_amd import ops
    rng = np.random.default_rng(41)
    n, k = 150_000, 128
    W = rng.normal(0, 0.05, (n, 128)).astype(np.float32)
    dup = [7, 2400, 2343, 2344, 70_000, 149_999]          # 150000/64 slices = 2344 keys each
    for d in dup[1:]:
        W[d] = W[dup[0]]
    Wh = ops.rownorm(torch.from_numpy(W))
    keep = rng.random(n) > 0.2
    keep[dup] = True
    keep[2343] = False
    for queries in ([7], [2400, 5, 149_999]):
        for kp in (None, keep):
            idx, sim = ops.cosine_topk(Wh, queries, k, keep=None if kp is None else kp.astype(np.uint8))
            idx, sim = idx.cpu().numpy(), sim.cpu().numpy()
            for j, q in enumerate(queries):
                s = ops.cosine_scores(Wh, q).cpu().numpy()
                oi, os_ = orc.topk_desc(s, k, exclude=q, mask=kp)
                assert (idx[j] == oi).all(), (queries, j)
                assert (sim[j] == os_).all()
    # the duplicates of the query are its nearest neighbours, in ascending index order
    idx, _ = ops.cosine_topk(Wh, [7], 10)
    assert idx.cpu().numpy()[0][:5].tolist() == [2343, 2344, 2400, 70_000, 149_999]


@pytest.mark.parametrize("waves", ["4", "8"])
def test_mfma_paths_on_random_odd_shapes(waves, monkeypatch):
    """Both workgroup shapes of k_cand (128- and 256-row) on sizes that are multiples of nothing: key
    tables ending inside a tile, query counts ending inside a workgroup / a wave, k from 1 to 127, key
    masks, and the masked predict variant — always identical to the exact kernels."""
    from anime_recommendations_amd import ops
    monkeypatch.setenv("ANIREC_TOPK_WAVES", waves)
    rng = np.random.default_rng(100 + int(waves))
    for trial in range(10):
        n = int(rng.integers(130, 9000))
        nq = int(rng.integers(1, 700))
        k = int(rng.choice([1, 2, 10, 37, 100, 127]))
        W = rng.normal(0, 0.05, (n, 128)).astype(np.float32)
        W[rng.integers(0, n, 3)] = W[0]                      # a few exact ties
        Wh = ops.rownorm(torch.from_numpy(W))
        q = rng.integers(0, n, nq).astype(np.int32)
        keep = (rng.random(n) > 0.3).astype(np.uint8) if trial % 2 else None
        ex = bool(trial % 3)
        ei, es = ops.cosine_topk(Wh, q, k, exclude_self=ex, keep=keep)
        mi, ms, _ = ops.cosine_topk_mfma(Wh, q, k, exclude_self=ex, keep=keep)
        assert torch.equal(mi, ei), (trial, n, nq, k)
        assert torch.equal(torch.nan_to_num(ms, nan=-9.0), torch.nan_to_num(es, nan=-9.0)), (trial, n, nq, k)
        # masked predict variant on the same odd shapes (keys = n "anime", queries = nq "users")
        if k <= 100:
            U = torch.from_numpy(rng.normal(0, 0.05, (max(nq, 2), 128)).astype(np.float32)).cuda()
            head = dict(w=1.1, b=0.05, gamma=0.95, beta=-0.1, mov_mean=0.02, mov_var=0.6)
            users = np.arange(nq, dtype=np.int32)
            _, bits = _watched_bits(rng, nq, n, 0.25)
            pe = ops.predict_topk(U, torch.from_numpy(W).cuda(), head, users, k, bits)
            pm = ops.predict_topk_mfma(U, torch.from_numpy(W).cuda(), head, users, k, bits)
            assert torch.equal(pm[0], pe[0]), (trial, n, nq, k)
            assert torch.equal(torch.nan_to_num(pm[1], nan=-9.0), torch.nan_to_num(pe[1], nan=-9.0))


def test_mfma_paths_on_tiny_tables():
    """Tables smaller than one key tile / one wave, k larger than the table: -1 / NaN padding identical to the
    exact kernels (rows with fewer than k candidates are flagged and re-run through them)."""
    from anime_recommendations_amd import ops
    rng = np.random.default_rng(77)
    for n, k in ((2, 1), (7, 10), (33, 40), (127, 127)):
        W = rng.normal(0, 0.05, (n, 128)).astype(np.float32)
        Wh = ops.rownorm(torch.from_numpy(W))
        q = np.arange(n, dtype=np.int32)
        ei, es = ops.cosine_topk(Wh, q, k)
        mi, ms, nfb = ops.cosine_topk_mfma(Wh, q, k)
        assert torch.equal(mi, ei), (n, k)
        assert torch.equal(torch.nan_to_num(ms, nan=-9.0), torch.nan_to_num(es, nan=-9.0)), (n, k)
        U = torch.from_numpy(rng.normal(0, 0.05, (n, 128)).astype(np.float32)).cuda()
        head = dict(w=1.1, b=0.05, gamma=0.95, beta=-0.1, mov_mean=0.02, mov_var=0.6)
        pe = ops.predict_topk(U, torch.from_numpy(W).cuda(), head, q, min(k, 100), None)
        pm = ops.predict_topk_mfma(U, torch.from_numpy(W).cuda(), head, q, min(k, 100), None)
        assert torch.equal(pm[0], pe[0]), (n, k)
        assert torch.equal(torch.nan_to_num(pm[1], nan=-9.0), torch.nan_to_num(pe[1], nan=-9.0)), (n, k)


@pytest.mark.parametrize("k", [10, 100])
def test_key_range_splits_give_the_same_lists(k, monkeypatch):
    """Small query sets share a row block's key tiles between up to 4 workgroups (own buffer regions, folded by
    k_refresh): the neighbour lists must not depend on the number of splits, and no row may need the fallback."""
    from anime_recommendations_amd import ops
    rng = np.random.default_rng(5)
    W = rng.normal(0, 0.05, (18_000, 128)).astype(np.float32)
    Wh = ops.rownorm(torch.from_numpy(W))
    q = torch.arange(3_000, dtype=torch.int32, device="cuda")
    ref = None
    for sp in ("1", "2", "3", "4"):
        monkeypatch.setenv("ANIREC_TOPK_SPLITS", sp)
        idx, sim, nfb = ops.cosine_topk_mfma(Wh, q, k)
        if sp != "2":                      # two splits of a k = 100 super-step may overflow their 88-entry regions
            assert nfb == 0, sp            # (those rows fall back to the exact path: still the same lists)
        if ref is None:
            ref = (idx.clone(), sim.clone())
        assert torch.equal(idx, ref[0]) and torch.equal(sim, ref[1]), sp
    ei, es = ops.cosine_topk(Wh, q[:256], k)
    assert torch.equal(ref[0][:256], ei) and torch.equal(ref[1][:256], es)


def test_threshold_prior_learnt_from_the_first_batch(monkeypatch):
    """Several query batches: the first runs without a prior, its k-th best scores give the others one.  Clustered rows
    (k-th neighbour at cosine ~0.9) set a high prior; the scattered rows in the later batch (k-th neighbour ~0.3)
    come out unproven under it and are re-run without — the lists must equal the exact path's, with and without."""
    from anime_recommendations_amd import ops
    rng = np.random.default_rng(12)
    n, k = 30_000, 50
    centres = rng.normal(0, 1, (60, 128))
    W = centres[rng.integers(0, 60, n)] + 0.25 * rng.normal(0, 1, (n, 128))
    W[-700:] = rng.normal(0, 1, (700, 128))                 # scattered rows, all in the second batch
    Wh = ops.rownorm(torch.from_numpy(W.astype(np.float32)))
    q = torch.arange(n, dtype=torch.int32, device="cuda")
    monkeypatch.setenv("ANIREC_TOPK_PRIOR", "0")
    i0, s0, _ = ops.cosine_topk_mfma(Wh, q, k, batch=16384)
    monkeypatch.setenv("ANIREC_TOPK_PRIOR", "1")
    i1, s1, _ = ops.cosine_topk_mfma(Wh, q, k, batch=16384)
    assert torch.equal(i0, i1) and torch.equal(s0, s1)
    probe = torch.cat([q[:300], q[16384:16684], q[-700:]])
    ei, es = ops.cosine_topk(Wh, probe, k)
    assert torch.equal(i1[probe.long()], ei) and torch.equal(s1[probe.long()], es)
    # an explicit prior above every score: every row is unproven, re-run, and still right
    i2, s2, _ = ops.cosine_topk_mfma(Wh, q[:2000], k, prior=0.999)
    assert torch.equal(i2, i0[:2000]) and torch.equal(s2, s0[:2000])


@pytest.mark.parametrize("nq", [49151, 49152])
def test_job_plan_boundary_learning_batch_on_and_off(nq):
    """49 152 queries is where the default plan starts to learn a prior from a 16 384-row first batch (below: one plain
    batch).  All-pairs over 49 152 keys, k = 40: lists equal the exact path's on rows of every batch."""
    from anime_recommendations_amd import ops
    g = torch.Generator(device="cuda")
    g.manual_seed(21)
    n, k = 49152, 40
    Wh = ops.rownorm(torch.randn(n, 128, generator=g, device="cuda") * 0.05)
    q = torch.arange(nq, dtype=torch.int32, device="cuda")
    stats = {}
    idx, sim, nfb = ops.cosine_topk_mfma(Wh, q, k, stats=stats)
    starts = stats["starts"]
    assert stats["learn_batches"] == (1 if nq >= 49152 else 0)
    assert starts == ([0, 16384, 49152] if nq >= 49152 else [0, 49151]) and nfb == 0
    probe = np.unique(np.concatenate([[s_, min(s_ + 1, nq - 1), max(s_ - 1, 0)] for s_ in starts[:-1]] +
                                     [np.arange(0, nq, 769), [nq - 1]]))
    probe = torch.from_numpy(probe).cuda()
    ei, es = ops.cosine_topk(Wh, probe.to(torch.int32), k)
    assert torch.equal(idx[probe], ei) and torch.equal(sim[probe], es)


@pytest.mark.parametrize("lanes", [1, 2, 3, 4])
def test_job_lanes_give_identical_lists(lanes):
    """The batches of a job are dealt to 1-4 interleaved stream-ordered chains (side streams forked / joined inside
    the call): results must not depend on it, run to run either."""
    from anime_recommendations_amd import ops
    rng = np.random.default_rng(31)
    n, k = 20_000, 30
    Wh = ops.rownorm(torch.from_numpy(rng.normal(0, 0.05, (n, 128)).astype(np.float32)))
    q = torch.from_numpy(rng.permutation(n).astype(np.int32)).cuda()
    ref = ops.cosine_topk_mfma(Wh, q, k, lanes=1, batch=20_000)
    for _ in range(2):
        stats = {}
        idx, sim, nfb = ops.cosine_topk_mfma(Wh, q, k, lanes=lanes, batch=2560, prior=None, stats=stats)
        assert 8 <= stats["batches"] <= 9 and stats["lanes"] == lanes
        assert torch.equal(idx, ref[0]) and torch.equal(sim, ref[1]) and nfb == ref[2]
    ei, es = ops.cosine_topk(Wh, q[:300], k)
    assert torch.equal(ref[0][:300], ei) and torch.equal(ref[1][:300], es)


@pytest.mark.parametrize("data", ["random", "clustered"])
@pytest.mark.parametrize("lanes", [1, 2])
def test_allpairs_symmetric_schedule_gives_the_plain_lists(data, lanes):
    """Every row a query, in order: a batch computes its dot products with the rows of later batches once for both
    sides (scores that reach the learnt prior also go to the later row's inbox) and skips the key tiles of earlier
    batches (include/anirec.h, prior_mode 3).  The lists must be those of the plain job and of the exact path on rows
    of every batch.  Clustered rows (260 members per cluster, cosine ~0.9 inside) push more pairs over the prior than
    an inbox holds: those rows are flagged, re-run without the shortcut, and still right.  A query list that is not
    the identity must not take the shortcut (checked on the device when forced)."""
    from anime_recommendations_amd import ops
    rng = np.random.default_rng(41)
    n, k = 52_000, 100
    if data == "random":
        W = rng.normal(0, 0.05, (n, 128))
    else:
        centres = rng.normal(0, 1, (200, 128))
        W = centres[rng.permutation(n) % 200] + 0.3 * rng.normal(0, 1, (n, 128))
    Wh = ops.rownorm(torch.from_numpy(W.astype(np.float32)))
    q = torch.arange(n, dtype=torch.int32, device="cuda")
    st0, st1 = {}, {}
    i0, s0, f0 = ops.cosine_topk_mfma(Wh, q, k, batch=16384, lanes=lanes, allpairs=False, stats=st0)
    i1, s1, f1 = ops.cosine_topk_mfma(Wh, q, k, batch=16384, lanes=lanes, allpairs=True, stats=st1)   # ("auto": from 196 608 rows)
    assert st0["allpairs"] is False and st1["allpairs"] is True and st1["batches"] == 4 and st1["learn_batches"] == 1
    assert torch.equal(i0, i1) and torch.equal(s0, s1)
    if data == "random":
        assert st1["rerun_rows"] <= 0.02 * n and f1 == 0
    starts = st1["starts"]
    probe = np.unique(np.concatenate([np.arange(s_, min(s_ + 100, n)) for s_ in starts[:-1]] +
                                     [np.arange(n - 100, n), np.arange(0, n, 523)]))
    probe = torch.from_numpy(probe).cuda()
    ei, es = ops.cosine_topk(Wh, probe.to(torch.int32), k)
    assert torch.equal(i1[probe], ei) and torch.equal(s1[probe], es)
    if data == "random" and lanes == 2:
        # forced on a permuted query list: the device check flags every row, the exact path answers
        qp = torch.from_numpy(rng.permutation(n).astype(np.int32)).cuda()
        st2 = {}
        i2, s2, f2 = ops.cosine_topk_mfma(Wh, qp, k, batch=16384, lanes=lanes, allpairs=True, stats=st2)
        assert st2["allpairs"] is True
        assert torch.equal(i2, i0[qp.long()]) and torch.equal(s2, s0[qp.long()])


def test_allpairs_equal_work_plan_with_a_ragged_last_batch():
    """The all-pairs plan proper (n >= 196 608: a learning batch + batches of equal work, the LAST one the largest)
    with a last batch of 256 m + 1..128 rows — its launches then have one 256-row workgroup more than ceil(rows / 128)
    / 2 (the chains' log buffers once were sized by the latter).  Lists equal the plain job's and the exact path's."""
    from anime_recommendations_amd import ops
    k = 40
    n = next(m for m in range(200_000, 201_000)
             if 0 < (m - ops.topk_allpairs_plan(m, k)[0][-2]) % 256 <= 128)
    starts, learn, _ = ops.topk_allpairs_plan(n, k)
    sizes = np.diff(starts)
    assert learn == 1 and len(starts) >= 4 and sizes[-1] == sizes.max() and all(s_ % 128 == 0 for s_ in starts[1:-1])
    g = torch.Generator(device="cuda")
    g.manual_seed(51)
    Wh = ops.rownorm(torch.randn(n, 128, generator=g, device="cuda") * 0.05)
    q = torch.arange(n, dtype=torch.int32, device="cuda")
    st = {}
    i1, s1, f1 = ops.cosine_topk_mfma(Wh, q, k, stats=st)
    assert st["allpairs"] is True and st["starts"] == starts and f1 == 0 and st["rerun_rows"] <= 0.02 * n
    i0, s0, _ = ops.cosine_topk_mfma(Wh, q, k, allpairs=False)
    assert torch.equal(i0, i1) and torch.equal(s0, s1)
    probe = torch.from_numpy(np.unique(np.concatenate([np.arange(s_, min(s_ + 64, n)) for s_ in starts[:-1]] +
                                                      [np.arange(n - 64, n)]))).cuda()
    ei, es = ops.cosine_topk(Wh, probe.to(torch.int32), k)
    assert torch.equal(i1[probe], ei) and torch.equal(s1[probe], es)


def test_allpairs_pilot_keeps_dense_tables_on_the_plain_job():
    """Rows with far more than k neighbours above the job's prior (clusters of 1 000 at k = 50) would overflow the
    all-pairs inboxes and be re-run: the pilot (512 sample rows against a 32 768-row sample) sees it and the job runs
    the plain schedule; a table of small clusters (150) takes the shortcut.  The lists are the exact path's either way."""
    from anime_recommendations_amd import ops
    n, k = 200_192, 50
    g = torch.Generator(device="cuda")
    g.manual_seed(61)
    q = torch.arange(n, dtype=torch.int32, device="cuda")
    for csize, expect in ((1000, False), (150, True)):
        centres = torch.randn(n // csize, 128, generator=g, device="cuda")
        assign = torch.randperm(n, generator=g, device="cuda") % (n // csize)
        Wh = ops.rownorm(centres[assign] + 0.3 * torch.randn(n, 128, generator=g, device="cuda"))
        st = {}
        idx, sim, _ = ops.cosine_topk_mfma(Wh, q, k, stats=st)
        assert st["pilot"]["allpairs"] is expect and st["allpairs"] is expect, st["pilot"]
        probe = torch.arange(0, n, 997, dtype=torch.int32, device="cuda")
        ei, es = ops.cosine_topk(Wh, probe, k)
        assert torch.equal(idx[probe.long()], ei) and torch.equal(sim[probe.long()], es)
