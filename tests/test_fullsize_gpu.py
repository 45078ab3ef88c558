"""Full BASELINE-size checks through size-independent properties (the oracle is too slow at
these sizes): determinism, exactness of the dense-only Adam update on untouched rows,
sortedness / self-exclusion / tie order of neighbour lists, agreement of the MFMA paths with the
exact kernels on samples."""
import numpy as np
import pytest
import torch

from oracle import anirec_oracle as orc

pytestmark = pytest.mark.gpu

N_USERS, N_ANIME, B = 350_000, 18_000, 10_000


def _train(steps, seed=1, lazy=None, use_graph=False, arena=8):
    import bench
    from anime_recommendations_amd.engine import TrainEngine
    from anime_recommendations_amd import schedule
    dev = torch.device("cuda:0")
    ui, ai, t = bench.synth_ratings(N_USERS, N_ANIME, steps * B, dev, seed=seed)
    U, A = bench.init_tables(N_USERS, N_ANIME, dev)
    eng = TrainEngine(N_USERS, N_ANIME, max_batch=B, arena_steps=arena, lazy=lazy)
    eng.set_head(w=1.2)
    eng.set_weights(U, A)
    eng.set_epoch(ui, ai, t, np.arange(steps) * B, np.full(steps, B), schedule.adam_alphas(1e-5, 1, steps))
    eng.run(steps, use_graph=use_graph)
    eng.synchronize()
    return eng, U.cpu().numpy(), ui.cpu().numpy()


def test_s109m_shape_training_is_deterministic_and_untouched_rows_are_bit_exact():
    eng1, U0, ui = _train(3)
    W1, M1, V1 = eng1.W.cpu().numpy(), eng1.M.cpu().numpy(), eng1.V.cpu().numpy()
    rec1 = eng1.read_state()
    assert (eng1.rowmap.cpu().numpy() == 0).all() and np.isfinite(rec1["last_loss"])
    eng1.close()
    eng2, _, _ = _train(3)
    assert (eng2.W.cpu().numpy() == W1).all() and (eng2.V.cpu().numpy() == V1).all()
    assert eng2.read_state()["loss_wsum"] == rec1["loss_wsum"]
    eng2.close()
    # rows no batch touched only see g = 2*l2*W: the fused kernel must equal the oracle's Adam bit for bit
    untouched = np.setdiff1d(np.arange(N_USERS), np.unique(ui))[:5000]
    w = U0[untouched].copy()
    m = np.zeros_like(w)
    v = np.zeros_like(w)
    for step in range(3):
        orc.adam_update(w, m, v, np.float32(2e-4) * w, orc.adam_alpha(1e-5, step + 1))
    assert (W1[untouched] == w).all() and (M1[untouched] == m).all() and (V1[untouched] == v).all()


def test_s109m_shape_lazy_adam_equals_dense_adam_bitwise():
    """BASELINE configs[2] table shape, 43 steps through the captured graph (five 8-step windows + an eager tail of
    three): the lazy dense Adam — the default at this size — leaves bit for bit the tables and Adam moments of the
    dense kernel (368 000 rows x 128, every element), the same scalar state, and a History loss within the rounding
    of its L2 sum."""
    dense, _, _ = _train(43, seed=5, lazy=False, use_graph=True, arena=32)
    lazy, _, _ = _train(43, seed=5, lazy=None, use_graph=True, arena=32)
    assert lazy.lazy and not dense.lazy
    assert torch.equal(dense.W, lazy.W) and torch.equal(dense.M, lazy.M) and torch.equal(dense.V, lazy.V)
    rd, rz = dense.read_state(), lazy.read_state()
    for k in ("w", "b", "gamma", "beta", "mov_mean", "mov_var", "last_mse", "bce_wsum", "se_sum", "step_fwd"):
        assert rd[k] == rz[k], k
    assert abs(float(rd["last_loss"]) - float(rz["last_loss"])) < 3e-6 * abs(float(rd["last_loss"]))
    assert abs(float(rd["loss_wsum"]) - float(rz["loss_wsum"])) < 3e-6 * abs(float(rd["loss_wsum"]))
    dense.close()
    lazy.close()


def _train_s7m(steps, use_graph):
    import bench
    from anime_recommendations_amd.engine import TrainEngine
    from anime_recommendations_amd import schedule
    n_u, n_a = bench.WORKLOADS["s7m"]
    dev = torch.device("cuda:0")
    ui, ai, t = bench.synth_ratings(n_u, n_a, steps * B, dev, seed=3)
    U, A = bench.init_tables(n_u, n_a, dev)
    eng = TrainEngine(n_u, n_a, max_batch=B, arena_steps=64)
    eng.set_head(w=1.2)
    eng.set_weights(U, A)
    eng.set_epoch(ui, ai, t, np.arange(steps) * B, np.full(steps, B), schedule.adam_alphas(1e-5, 1, steps))
    eng.run(steps, use_graph=use_graph)
    eng.synchronize()
    out = (eng.W.cpu().numpy(), eng.M.cpu().numpy(), eng.V.cpu().numpy(), eng.read_state(),
           eng.rowmap.cpu().numpy())
    eng.close()
    return out, A.cpu().numpy(), ai.cpu().numpy()


def test_s7m_shape_graph_is_bitwise_the_eager_run_and_untouched_rows_are_bit_exact():
    """BASELINE configs[1] table shape (15 000 x 17 560, B = 10 000): the captured-graph loop (two 32-step replays +
    a 6-step eager tail) against the eager loop, bit for bit, and the dense-only Adam update of rows no batch
    touched against the oracle."""
    steps = 70
    (W0, M0, V0, rec0, rm0), A0, ai = _train_s7m(steps, use_graph=False)
    assert (rm0 == 0).all() and np.isfinite(rec0["last_loss"]) and int(rec0["step_fwd"]) == steps
    for _ in range(2):
        (W1, M1, V1, rec1, rm1), _, _ = _train_s7m(steps, use_graph=True)
        assert (W1 == W0).all() and (M1 == M0).all() and (V1 == V0).all() and (rm1 == 0).all()
        assert rec1["loss_wsum"] == rec0["loss_wsum"] and rec1["w"] == rec0["w"] and rec1["mov_var"] == rec0["mov_var"]
    # anime rows no batch touched only see g = 2*l2*W: bit for bit the oracle's Adam
    n_u = 15_000
    untouched = np.setdiff1d(np.arange(A0.shape[0]), np.unique(ai))[:3000]
    assert len(untouched) > 20
    w = A0[untouched].copy()
    m = np.zeros_like(w)
    v = np.zeros_like(w)
    for step in range(steps):
        orc.adam_update(w, m, v, np.float32(2e-4) * w, orc.adam_alpha(1e-5, step + 1))
    assert (W0[n_u + untouched] == w).all() and (M0[n_u + untouched] == m).all() and (V0[n_u + untouched] == v).all()


def test_350k_neighbour_lists_properties_and_sample_equals_exact_path():
    from anime_recommendations_amd import ops
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    W = torch.randn(N_USERS, 128, generator=g, device="cuda") * 0.05
    W[1000] = W[5]                                     # an exact duplicate pair
    Wh = ops.rownorm(W)
    q = torch.arange(0, 4096, dtype=torch.int32, device="cuda")
    idx, sim, nfb = ops.cosine_topk_mfma(Wh, q, 100)
    idx, sim = idx.cpu().numpy(), sim.cpu().numpy()
    assert nfb == 0
    assert (idx >= 0).all() and (idx != np.arange(4096)[:, None]).all()            # k rows, query dropped
    assert (np.diff(sim, axis=1) <= 0).all()                                        # descending
    tie = np.diff(sim, axis=1) == 0
    assert (np.diff(idx, axis=1)[tie] > 0).all()                                    # ties -> ascending index
    assert idx[5, 0] == 1000                                                        # the duplicate is the nearest
    assert all(len(set(r)) == 100 for r in idx[:256])                               # no repeats
    sample = torch.tensor([0, 5, 17, 999, 4095], dtype=torch.int32, device="cuda")
    ei, es = ops.cosine_topk(Wh, sample, 100)
    assert (ei.cpu().numpy() == idx[[0, 5, 17, 999, 4095]]).all()
    assert (es.cpu().numpy() == sim[[0, 5, 17, 999, 4095]]).all()


def test_anime_18k_allpairs_top100_all_queries_equal_exact_path():
    """BASELINE configs[3], anime leg, exactly as bench.py runs it: ALL 18 000 rows as queries against all 18 000
    keys, k = 100 (the plan this size takes: one batch, 128-row workgroups, key-range splits) — every list has k
    distinct neighbours in descending order without the query, no row falls back, and 640 rows (first, last, a
    planted duplicate pair, a random sample) equal the exact kernels' lists and scores bit for bit."""
    from anime_recommendations_amd import ops
    n = 18_000
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    W = torch.randn(n, 128, generator=g, device="cuda") * 0.05
    W[7777] = W[42]                                    # an exact duplicate pair: score 1, tie order by index
    Wh = ops.rownorm(W)
    q = torch.arange(n, dtype=torch.int32, device="cuda")
    stats = {}
    idx, sim, nfb = ops.cosine_topk_mfma(Wh, q, 100, stats=stats)
    assert nfb == 0 and stats["batches"] == 1
    idx_h, sim_h = idx.cpu().numpy(), sim.cpu().numpy()
    assert (idx_h >= 0).all() and (idx_h != np.arange(n)[:, None]).all()
    assert (np.diff(sim_h, axis=1) <= 0).all()
    tie = np.diff(sim_h, axis=1) == 0
    assert (np.diff(idx_h, axis=1)[tie] > 0).all()
    assert idx_h[42, 0] == 7777 and idx_h[7777, 0] == 42
    assert (np.sort(idx_h, axis=1)[:, 1:] != np.sort(idx_h, axis=1)[:, :-1]).all()          # no repeats in any row
    rng = np.random.default_rng(5)
    rows = np.unique(np.concatenate([[0, 42, 7777, n - 1], rng.choice(n, 640, replace=False)]))
    sample = torch.from_numpy(rows.astype(np.int32)).cuda()
    ei, es = ops.cosine_topk(Wh, sample, 100)
    assert (ei.cpu().numpy() == idx_h[rows]).all()
    assert (es.cpu().numpy() == sim_h[rows]).all()


def test_350k_allpairs_top100_rows_of_every_batch_equal_the_exact_path():
    """BASELINE configs[3] users job exactly as bench.py runs it (350 k x 350 k, k = 100, the library's all-pairs plan:
    a 16 384-row learning batch, then batches of equal work on two interleaved chains, each computing its dot products
    with the rows of later batches once for both sides; similar_users.py:290-312 for every user).
    Rows drawn from EVERY batch — first / last row of each, a random sample of each, and rows that come out unproven
    under the learnt prior and are re-run without it — must equal the exact kernels' lists bit for bit.
    The bulk of the table lives (almost) in a 64-dimensional subspace (k-th best cosine ~0.43); 360 planted rows are
    isotropic in all 128 dimensions (k-th best ~0.30, far below the learnt prior): those are the re-run rows."""
    from anime_recommendations_amd import ops
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    n, k = N_USERS, 100
    W = torch.randn(n, 128, generator=g, device="cuda") * 0.05
    W[:, 64:] *= 0.04
    st0, learn0, _ = ops.topk_allpairs_plan(n, k)
    assert learn0 == 1 and st0[1] == 16384 and len(st0) >= 5
    planted = torch.cat([torch.arange(100, 140), torch.arange(st0[1] + 7, st0[1] + 87),
                         torch.arange(st0[2] - 40, st0[2] + 40), torch.arange(st0[3] + 1000, st0[3] + 1080),
                         torch.arange(n - 80, n)]).cuda()
    W[planted] = torch.randn(planted.numel(), 128, generator=g, device="cuda") * 0.05
    Wh = ops.rownorm(W)
    q = torch.arange(n, dtype=torch.int32, device="cuda")
    stats = {}
    idx, sim, nfb = ops.cosine_topk_mfma(Wh, q, k, stats=stats)
    starts = stats["starts"]
    assert starts == st0 and stats["learn_batches"] == 1 and stats["lanes"] == 2 and stats["allpairs"] is True
    late = planted[planted >= starts[1]]
    assert stats["rerun_rows"] >= late.numel()            # every planted row behind the learning batch was refuted
    assert stats["rerun_rows"] < 2000 and nfb <= 4
    rng = np.random.default_rng(5)
    probe = [planted.cpu().numpy()]
    for b in range(len(starts) - 1):
        lo, hi = starts[b], starts[b + 1]
        probe.append(np.array([lo, lo + 1, hi - 2, hi - 1]))
        probe.append(rng.integers(lo, hi, 96))
    probe = torch.from_numpy(np.unique(np.concatenate(probe))).cuda()
    ei, es = ops.cosine_topk(Wh, probe.to(torch.int32), k)
    assert torch.equal(idx[probe], ei) and torch.equal(sim[probe], es)
    # whole-job properties
    assert bool((idx >= 0).all()) and bool((idx != q[:, None]).all())
    assert bool((sim[:, 1:] <= sim[:, :-1]).all())
    # the plain job (every batch on the whole key stream, the default plan): the same lists, row for row
    st2 = {}
    i2, s2, _ = ops.cosine_topk_mfma(Wh, q, k, allpairs=False, stats=st2)
    assert st2["allpairs"] is False and st2["starts"] == ops.topk_job_plan(n, k)[0]
    assert torch.equal(i2, idx) and torch.equal(s2, sim)
    del i2, s2
    # one chain instead of two, and no prior at all: the same lists
    i1, s1, _ = ops.cosine_topk_mfma(Wh, q[: starts[2]], k, lanes=1)
    assert torch.equal(i1, idx[: starts[2]]) and torch.equal(s1, sim[: starts[2]])


def test_predict_grid_full_anime_table_matches_pairwise_kernel():
    from anime_recommendations_amd import ops
    g = torch.Generator(device="cuda")
    g.manual_seed(4)
    U = torch.randn(N_USERS, 128, generator=g, device="cuda") * 0.05
    A = torch.randn(N_ANIME, 128, generator=g, device="cuda") * 0.05
    head = dict(w=1.3, b=0.1, gamma=0.9, beta=-0.2, mov_mean=0.05, mov_var=0.4)
    users = torch.arange(1000, 3048, dtype=torch.int32, device="cuda")
    G = ops.predict_grid_mfma(U, A, head, users)
    assert G.shape == (2048, N_ANIME) and bool(((G > 0) & (G < 1)).all())
    rng = np.random.default_rng(0)
    ju, ja = rng.integers(0, 2048, 20000), rng.integers(0, N_ANIME, 20000)
    p = ops.predict_pairs(U, A, head, users[torch.from_numpy(ju).cuda()], ja)
    np.testing.assert_allclose(G.cpu().numpy()[ju, ja], p.cpu().numpy(), atol=3e-6)


def test_predict_topk_mfma_c5_sample_equals_exact_path():
    """BASELINE C5 (100 k users x 18 k anime, ~25 % watched): the MFMA path on all users, the exact
    kernels on a sample of them — identical lists; hardly any fallback."""
    from anime_recommendations_amd import ops
    n_u, n_a, nq, k = 350_000, 18_000, 100_000, 10
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    U = torch.randn(n_u, 128, generator=g, device="cuda") * 0.05
    A = torch.randn(n_a, 128, generator=g, device="cuda") * 0.05
    head = dict(w=1.3, b=0.1, gamma=0.9, beta=-0.2, mov_mean=0.05, mov_var=0.4)
    users = torch.arange(nq, dtype=torch.int32, device="cuda")
    shape = (nq, (n_a + 31) // 32)
    watched = torch.randint(-2 ** 31, 2 ** 31 - 1, shape, generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
    watched &= torch.randint(-2 ** 31, 2 ** 31 - 1, shape, generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
    mi, mp, nfb = ops.predict_topk_mfma(U, A, head, users, k, watched)
    assert nfb <= nq // 1000
    sample = torch.arange(0, nq, 97, device="cuda")
    ei, ep = ops.predict_topk(U, A, head, users[sample], k, watched[sample])
    assert torch.equal(mi[sample], ei) and torch.equal(mp[sample], ep)
    # every recommended anime is unwatched, ratings are sorted
    idx = mi.long()
    bit = (watched.gather(1, idx >> 5) >> (idx & 31)) & 1
    assert int(bit.sum()) == 0
    assert bool((mp[:, 1:] <= mp[:, :-1]).all())
