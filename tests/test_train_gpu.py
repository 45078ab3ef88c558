"""GPU parity tests of the training hot path: libanirec (through its C ABI) vs the CPU
oracle on the same seeded inputs.  Tolerances: weights/ratings are fp32 — the bar from
BASELINE.json is 1e-5 on ratings; summation order differs between the HIP kernels
(butterfly / chunked) and NumPy (pairwise), so intermediate fp32 values are compared at a
few ulp, integer/index results exactly."""
import numpy as np
import pytest
import torch

from oracle import anirec_oracle as orc

pytestmark = pytest.mark.gpu


def _problem(seed, n_u, n_a, n, zipf=1.2):
    rng = np.random.default_rng(seed)
    U = rng.uniform(-0.05, 0.05, (n_u, 128)).astype(np.float32)
    A = rng.uniform(-0.05, 0.05, (n_a, 128)).astype(np.float32)
    ui = rng.integers(0, n_u, n).astype(np.int64)
    ai = ((rng.zipf(zipf, n) - 1) % n_a).astype(np.int64)
    t = (rng.integers(0, 11, n) / 10).astype(np.float32)
    return U, A, ui, ai, t


def _engine(U, A, B, head_w=1.2, arena=8):
    from anime_recommendations_amd.engine import TrainEngine
    eng = TrainEngine(U.shape[0], A.shape[0], max_batch=B, arena_steps=arena)
    eng.set_head(w=head_w)
    eng.set_weights(U, A)
    return eng


def _schedule(n, B, lr, t0=0):
    starts = np.arange(0, n, B)
    counts = np.minimum(B, n - starts)
    alphas = [orc.adam_alpha(lr, t0 + i + 1) for i in range(len(starts))]
    return starts, counts, alphas


def test_prep_sorts_batches_and_cuts_chunks():
    from anime_recommendations_amd.engine import read_slot
    U, A, ui, ai, t = _problem(1, 5000, 700, 3 * 4000 - 123, zipf=1.1)
    B = 4000
    eng = _engine(U, A, B)
    starts, counts, alphas = _schedule(len(ui), B, 1e-5)
    eng.set_epoch(ui, ai, t, starts, counts, alphas)
    eng.prep(0, len(starts))
    n_u = U.shape[0]
    for s, (st, nb) in enumerate(zip(starts, counts)):
        nch, sidx, oth, chunks = read_slot(eng, s)
        for T, (key, other, koff, ooff) in enumerate([(ui, ai, 0, n_u), (ai, ui, n_u, 0)]):
            k = key[st:st + nb]
            order = np.argsort(k, kind="stable")
            assert (sidx[T, :nb] == order).all(), (s, T)
            assert (oth[T, :nb] == other[st:st + nb][order] + ooff).all()
            # chunks tile [0, nb) without crossing a row boundary, <= 32 long
            ch = chunks[T, :nch[T]]
            assert ch[0, 1] == 0 and (ch[:, 1] + ch[:, 2])[-1] == nb
            assert ((ch[:-1, 1] + ch[:-1, 2]) == ch[1:, 1]).all()
            assert ch[:, 2].max() <= 32 and ch[:, 2].min() >= 1
            ks = k[order]
            for row, start, ln, first in ch:
                assert (ks[start:start + ln] + koff == row).all()
            # first-chunk flag carries the number of chunks of that row
            firsts = ch[ch[:, 3] > 0]
            rows, cnts = np.unique(ch[:, 0], return_counts=True)
            assert (np.sort(firsts[:, 0]) == rows).all()
            assert dict(zip(firsts[:, 0], firsts[:, 3])) == dict(zip(rows, cnts))
    eng.close()


def test_forward_and_head_match_oracle():
    from anime_recommendations_amd.engine import read_ws
    U, A, ui, ai, t = _problem(2, 3000, 500, 2500)
    B = 2500
    eng = _engine(U, A, B)
    eng.set_epoch(ui, ai, t, [0], [B], [orc.adam_alpha(1e-5, 1)])
    eng.fwd()
    head = orc.new_head(w=1.2)
    f, g, met = orc.grads(U, A, ui, ai, t, head)
    eng.synchronize()
    pk = eng.packets.cpu().numpy()
    pc = (B + 3) & ~3
    np.testing.assert_allclose(pk[:B], f["c"], atol=3e-7)
    np.testing.assert_array_equal(pk[pc:pc + B], t)
    assert pk[2 * pc:2 * pc + 1].view(np.int32)[0] == B
    np.testing.assert_allclose(read_ws(eng, "su")[:B], f["su"], rtol=1e-6)
    np.testing.assert_allclose(read_ws(eng, "sa")[:B], f["sa"], rtol=1e-6)
    eng.head()
    dy = read_ws(eng, "dy")[:B]
    dy_o = (f["p"] - t) / np.float32(B)
    np.testing.assert_allclose(dy, dy_o, atol=np.abs(dy_o).max() * 2e-5)
    nblk = (B + 255) // 256
    hp = read_ws(eng, "hpart")[:nblk * 8].reshape(nblk, 8).astype(np.float64).sum(0)
    zh = (f["z"] - f["mu"]) * f["r"]
    assert abs(hp[0] - float(np.sum(dy_o, dtype=np.float64))) < 1e-7            # d beta
    assert abs(hp[1] - float(np.sum(dy_o.astype(np.float64) * zh))) < 1e-7      # d gamma
    assert abs(hp[2] / B - float(met["bce"])) < 2e-6
    assert abs(hp[3] / B - float(met["mse"])) < 1e-6
    pub = np.frombuffer(read_ws(eng, "pub", np.uint8).tobytes()[:56],
                        dtype=[("i", "<i4", (4,)), ("f", "<f4", (10,))])[0]
    assert list(pub["i"]) == [0, B, nblk, 0]
    assert abs(pub["f"][1] - f["mu"]) < 1e-6 and abs(pub["f"][2] - f["var"]) < 1e-7
    assert abs(pub["f"][8] - 1e-4) < 1e-10        # lambda rides in the step constants; the L2 SUM is taken at the finish
    # bwd + adam finish the step: scalar state and History metrics
    eng.prep(0, 1)
    eng.bwd()
    eng.adam()
    rec = eng.read_state()
    assert abs(rec["bn_mu"] - f["mu"]) < 1e-6 and abs(rec["bn_var"] - f["var"]) < 1e-7
    assert abs(rec["last_loss"] - met["loss"]) < 2e-6
    assert abs(rec["last_mse"] - met["mse"]) < 1e-6
    assert abs(rec["reg_sumsq"] - met["reg"]) / met["reg"] < 1e-6
    assert rec["step_fwd"] == 1 and rec["step_bwd"] == 0
    # first Adam step on the scalars moves each by lr*sign(g) (up to epsilon)
    assert abs((float(rec["w"]) - 1.2) + np.sign(float(g["w"])) * 1e-5) < 5e-7
    assert abs((float(rec["gamma"]) - 1.0) + np.sign(float(g["gamma"])) * 1e-5) < 5e-7
    eng.close()


@pytest.mark.parametrize("n_u,n_a,B,steps,zipf", [(300, 200, 256, 1, 1.3), (4000, 900, 1000, 10, 1.15),
                                                   (2000, 64, 4096, 3, 1.05)])
def test_train_steps_match_oracle(n_u, n_a, B, steps, zipf):
    n = B * steps - (B // 3 if steps > 1 else 0)      # ragged last batch
    U, A, ui, ai, t = _problem(3, n_u, n_a, n, zipf)
    lr = 3e-5
    st = orc.new_state(U, A, orc.new_head(w=1.2))
    starts, counts, alphas = _schedule(n, B, lr)
    mets = [orc.train_step(st, ui[s:s + c], ai[s:s + c], t[s:s + c], lr)[0] for s, c in zip(starts, counts)]
    eng = _engine(U, A, B)
    eng.set_epoch(ui, ai, t, starts, counts, alphas)
    eng.run(len(starts), use_graph=False)
    rec = eng.read_state()
    assert rec["step_fwd"] == len(starts)
    # weights move by ~lr per step: compare at 1e-3 of one step
    tol = lr * 2e-3 * len(starts) + 1e-9
    np.testing.assert_allclose(eng.U.cpu().numpy(), st["U"], atol=tol)
    np.testing.assert_allclose(eng.A.cpu().numpy(), st["A"], atol=tol)
    M = eng.M.cpu().numpy()
    np.testing.assert_allclose(M[:n_u], st["mU"], atol=np.abs(st["mU"]).max() * 1e-4)
    np.testing.assert_allclose(M[n_u:], st["mA"], atol=np.abs(st["mA"]).max() * 1e-4)
    V = eng.V.cpu().numpy()
    np.testing.assert_allclose(V[n_u:], st["vA"], atol=np.abs(st["vA"]).max() * 1e-4)
    h = st["head"]
    for k in ("w", "gamma", "beta"):
        assert abs(float(rec[k]) - float(h[k])) < tol, k
    # d loss/d b == 0 analytically (BatchNorm removes the mean): its Adam step is driven by
    # rounding noise in BOTH implementations, bounded by lr per step; it cannot change any output
    assert abs(float(rec["b"]) - float(h["b"])) <= 2.05 * lr * len(starts)
    assert abs(rec["mov_mean"] - h["mov_mean"]) < 1e-6 and abs(rec["mov_var"] - h["mov_var"]) < 1e-6
    assert abs(rec["last_loss"] - mets[-1]["loss"]) < 5e-6
    loss_epoch = sum(float(m["loss"]) * c for m, c in zip(mets, counts)) / n
    assert abs(eng.epoch_metrics()[0] - loss_epoch) < 5e-6
    assert (eng.rowmap.cpu().numpy() == 0).all()       # adam leaves the row map clean
    # predicted ratings after training: the BASELINE bar (1e-5)
    from anime_recommendations_amd import ops
    hd = {k: float(rec[k]) for k in ("w", "b", "gamma", "beta", "mov_mean", "mov_var")}
    p = ops.predict_pairs(eng.U, eng.A, hd, ui[:500], ai[:500]).cpu().numpy()
    po = orc.predict_pairs(st["U"], st["A"], h, ui[:500], ai[:500])
    np.testing.assert_allclose(p, po, atol=1e-5)
    eng.close()


def test_graph_replay_is_bitwise_identical_and_deterministic():
    U, A, ui, ai, t = _problem(4, 6000, 800, 40 * 1000, 1.1)
    B, lr = 1000, 5e-5
    starts, counts, alphas = _schedule(len(ui), B, lr)
    outs = []
    for use_graph in (False, True, True):
        eng = _engine(U, A, B, arena=16)
        eng.set_epoch(ui, ai, t, starts, counts, alphas)
        eng.run(len(starts), use_graph=use_graph)
        eng.synchronize()
        outs.append((eng.W.cpu().numpy().copy(), eng.V.cpu().numpy().copy(), eng.read_state()))
        eng.close()
    for W, V, rec in outs[1:]:
        assert (W == outs[0][0]).all() and (V == outs[0][1]).all()
        assert rec["loss_wsum"] == outs[0][2]["loss_wsum"] and rec["w"] == outs[0][2]["w"]


def test_adam_flat_is_bit_exact():
    from anime_recommendations_amd import ops
    rng = np.random.default_rng(5)
    n = 1 << 18
    w = rng.normal(0, 0.05, n).astype(np.float32)
    m = rng.normal(0, 1e-5, n).astype(np.float32)
    v = (rng.normal(0, 1e-5, n) ** 2).astype(np.float32)
    g = rng.normal(0, 1e-4, n).astype(np.float32)
    g[:100] = 0
    alpha = orc.adam_alpha(4.2e-5, 1234)
    tw, tm, tv, tg = (torch.from_numpy(x.copy()).cuda() for x in (w, m, v, g))
    ops.adam_flat(tw, tm, tv, tg, alpha)
    torch.cuda.synchronize()
    orc.adam_update(w, m, v, g, alpha)
    assert (tw.cpu().numpy() == w).all() and (tm.cpu().numpy() == m).all() and (tv.cpu().numpy() == v).all()


def test_evaluate_matches_oracle():
    U, A, ui, ai, t = _problem(6, 1500, 400, 3000)
    st = orc.new_state(U, A, orc.new_head(w=1.3, b=0.1, gamma=0.9, beta=-0.2, mov_mean=0.05, mov_var=0.4))
    eng = _engine(U, A, 1024)
    eng.set_head(w=1.3, b=0.1, gamma=0.9, beta=-0.2, mov_mean=0.05, mov_var=0.4)
    vl, vm = eng.evaluate(ui, ai, t)
    ev = orc.evaluate(st, ui, ai, t)
    assert abs(vl - float(ev["val_loss"])) < 3e-6 and abs(vm - float(ev["val_mse"])) < 1e-6
    eng.close()


def test_zero_and_duplicate_rows_edge_cases():
    U, A, ui, ai, t = _problem(7, 50, 20, 512)
    U[3] = 0.0                      # zero user row: l2_normalize clamps, gradient path gated
    ai[:300] = 5                    # one anime row with 300 contributions (10 chunks)
    ui[:40] = 3
    st = orc.new_state(U, A, orc.new_head(w=0.8))
    lr = 1e-5
    orc.train_step(st, ui, ai, t, lr)
    eng = _engine(U, A, 512, head_w=0.8)
    eng.set_epoch(ui, ai, t, [0], [512], [orc.adam_alpha(lr, 1)])
    eng.run(1, use_graph=False)
    eng.synchronize()               # the engine works on its own stream
    np.testing.assert_allclose(eng.U.cpu().numpy(), st["U"], atol=3e-8)
    np.testing.assert_allclose(eng.A.cpu().numpy(), st["A"], atol=3e-8)
    eng.close()


def test_gather_ratings_matches_numpy():
    from anime_recommendations_amd import ops
    rng = np.random.default_rng(8)
    n = 100003
    u = rng.integers(0, 1000, n).astype(np.int32)
    a = rng.integers(0, 1000, n).astype(np.int32)
    t = rng.random(n).astype(np.float32)
    perm = rng.permutation(n)
    uo, ao, to = ops.gather_ratings(torch.from_numpy(u).cuda(), torch.from_numpy(a).cuda(),
                                    torch.from_numpy(t).cuda(), torch.from_numpy(perm).cuda())
    assert (uo.cpu().numpy() == u[perm]).all() and (ao.cpu().numpy() == a[perm]).all()
    assert (to.cpu().numpy() == t[perm]).all()


def test_long_horizon_graph_run_tracks_the_c_oracle():
    """300 steps through the hipGraph path (arena refills, several graph replays, a learning rate that
    changes between 'epochs') against the C restatement of the Keras step: the two fp32 trajectories
    must stay together — per-step drift is rounding noise, it must not compound."""
    from anime_recommendations_amd import ops, schedule
    from oracle import c_oracle
    n_u, n_a, B, steps = 3000, 700, 1024, 300
    U, A, ui, ai, t = _problem(9, n_u, n_a, B * steps, 1.1)
    lrs = [1e-4, 3e-4, 2e-4]                                   # three "epochs" of 100 steps
    alphas = np.concatenate([schedule.adam_alphas(lr, 1 + 100 * e, 100) for e, lr in enumerate(lrs)])
    starts = np.arange(steps) * B
    counts = np.full(steps, B)
    st = orc.new_state(U, A, orc.new_head(w=1.2))
    st["head"]["m"] = np.zeros(4, np.float32)
    st["head"]["v"] = np.zeros(4, np.float32)
    # the oracle step by step (bit-identical to one 300-step call: every bit of state is passed in and out), so that
    # the Dense bias after every step is on record: its per-step rms move is the scale of the noise walk below
    b_path = [float(st["head"]["b"])]
    for s in range(steps):
        sl = slice(s * B, (s + 1) * B)
        met = c_oracle.train_run(st, ui[sl], ai[sl], t[sl], B, alphas[s:s + 1])
        b_path.append(float(st["head"]["b"]))
    sigma_b = float(np.sqrt(np.mean(np.diff(b_path) ** 2)))
    eng = _engine(U, A, B)
    eng.set_epoch(ui, ai, t, starts, counts, alphas)
    eng.run(steps, use_graph=True)
    rec = eng.read_state()
    assert rec["step_fwd"] == steps
    moved = np.abs(st["U"] - U).max()
    assert moved > 50 * max(lrs) * 0.5                          # the weights really travelled
    # after 300 steps of ~lr each the two runs agree to a small fraction of ONE step
    np.testing.assert_allclose(eng.U.cpu().numpy(), st["U"], atol=max(lrs) * 0.05)
    np.testing.assert_allclose(eng.A.cpu().numpy(), st["A"], atol=max(lrs) * 0.05)
    assert abs(rec["last_loss"] - met["loss"]) < 2e-5
    # d loss/d b == 0 analytically, so the Dense bias random-walks on rounding noise in both runs (bounded by
    # lr per step) and drags the moving mean of z = w c + b along; only b - mov_mean reaches any output
    h = st["head"]
    assert abs(float(rec["b"]) - float(h["b"])) <= 2.05 * float(np.sum(lrs)) * 100
    # What IS asserted about that combination: D = (b - mov_mean)_gpu - (b - mov_mean)_oracle is the difference of the
    # two bias walks passed through the moving mean's filter, D_t = sum_s 0.99^(t-s) xi_s, xi = the difference of the
    # two runs' bias moves (per-step scale sqrt(2) sigma_b, sigma_b taken from the oracle's own recorded path above:
    # ~0.02 lr, the Adam epsilon damps a gradient of ~1e-7).  Adam's first moment correlates the moves over ~10 steps,
    # which inflates the sum by at most sqrt((1 + beta1) / (1 - beta1)):
    #     std(D) <= sigma_b * sqrt(2 / (1 - 0.99^2)) * sqrt(1.9 / 0.1) = 43.7 sigma_b     (2.7e-4 here;
    # round 3 observed 1.5e-4 typically and 3.0e-4 once, against a measured-once constant of 2e-4).
    # The bar is five of those standard deviations.  A bias gradient that is NOT noise (a wrong d loss / d b) moves b
    # by ~alpha every step and D settles near 99 alpha ~ 1e-2, seven times the bar: that is what this line catches.
    gain = np.sqrt(2.0 / (1.0 - 0.99 ** 2)) * np.sqrt(1.9 / 0.1)
    d_walk = abs((float(rec["b"]) - float(rec["mov_mean"])) - (float(h["b"]) - float(h["mov_mean"])))
    assert 0.0 < sigma_b < 0.1 * max(lrs)                       # the bias really only moves on noise
    assert d_walk < 5.0 * gain * sigma_b, (d_walk, sigma_b)
    assert abs(rec["mov_var"] - h["mov_var"]) < 1e-5
    for k in ("w", "gamma", "beta"):
        assert abs(float(rec[k]) - float(h[k])) < max(lrs) * 0.05, k
    # Ratings: with each run's OWN (b, mov_mean) the inference outputs of two correct implementations differ by
    # the random walk above (measured here: 1.5e-4 after 300 steps, every rating shifted the same way) — the
    # reference is just as irreproducible against itself.  With that one noise-driven scalar pair taken from
    # the oracle, everything the data determined (both tables, w, gamma, beta, moving variance) meets the bar.
    hd = {k: float(rec[k]) for k in ("w", "gamma", "beta", "mov_var")}
    hd["b"], hd["mov_mean"] = float(h["b"]), float(h["mov_mean"])
    p = ops.predict_pairs(eng.U, eng.A, hd, ui[:2000], ai[:2000]).cpu().numpy()
    po = orc.predict_pairs(st["U"], st["A"], st["head"], ui[:2000], ai[:2000])
    np.testing.assert_allclose(p, po, atol=1e-5)               # BASELINE.json's bar on the ratings
    own = {k: float(rec[k]) for k in ("w", "b", "gamma", "beta", "mov_mean", "mov_var")}
    p_own = ops.predict_pairs(eng.U, eng.A, own, ui[:2000], ai[:2000]).cpu().numpy()
    # with each run's own pair the ratings differ by the walk only: |d sigmoid| <= 1/4 |d y|, d y = gamma * rs * D
    rs_inf = 1.0 / np.sqrt(float(rec["mov_var"]) + 1e-3)
    assert np.abs(p_own - po).max() < 0.25 * abs(float(rec["gamma"])) * rs_inf * 5.0 * gain * sigma_b + 1e-5
    eng.close()


def test_lazy_replay_square_root_is_sqrtf_on_every_float_of_its_range_and_divide_on_a_sample():
    """The lazy update replays Adam steps with a short correctly rounded square root (v_rsq_f32 + one Newton
    correction with an exact residual) and the compiler's divide chain without its scaling / fix-up steps.  The square root is checked against
    sqrtf on EVERY float of [2^-96, 2^96] (exhaustive: 1.6e9 inputs), the divide against IEEE `/` on 2^30 operand
    pairs of the admitted ranges: zero mismatches, i.e. inside its range test the replay performs exactly the dense
    kernel's fp32 operations (outside it, the dense kernel's own code runs)."""
    import ctypes as C
    from anime_recommendations_amd import _lib
    lib = _lib.load()
    cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
    _lib.check(lib.anirec_selftest_lazy_math(C.c_uint64(1 << 30), _lib.ptr(cnt), None), "anirec_selftest_lazy_math")
    torch.cuda.synchronize()
    assert cnt.tolist() == [0, 0], cnt.tolist()
    # the comparison itself can fail: the same code without its Newton correction (s0 = x * rsq(x)) is only faithful;
    # with a second correction it must stay exact
    import os
    left = {}
    for steps in ("0", "2"):
        os.environ["ANIREC_SELFTEST_NEWTON"] = steps
        try:
            _lib.check(lib.anirec_selftest_lazy_math(C.c_uint64(0), _lib.ptr(cnt), None), "anirec_selftest_lazy_math")
            torch.cuda.synchronize()
        finally:
            del os.environ["ANIREC_SELFTEST_NEWTON"]
        left[steps] = cnt.tolist()
    assert left["0"][0] > 1_000_000 and left["0"][1] == 0 and left["2"] == [0, 0], left


@pytest.mark.parametrize("use_graph", [False, True])
def test_lazy_adam_is_bitwise_the_dense_update(use_graph):
    """The lazy dense Adam (rows a batch does not touch take their L2-only steps later, several at a time: catch-up
    before fwd, sparse step after bwd, a flush every 8 steps) must leave exactly the tables, Adam moments and scalar
    state of the dense kernel — every element sees the same fp32 operations in the same order — through ragged run()
    calls, windows cut short, graph replays and their eager tails; the History loss agrees to the rounding of its
    L2 sum (another summation order)."""
    from anime_recommendations_amd.engine import TrainEngine
    n_u, n_a, B, steps = 5000, 900, 512, 77
    U, A, ui, ai, t = _problem(17, n_u, n_a, B * steps - 100, 1.1)
    # rows the replay's short arithmetic sequences are NOT exact for — exact zeros, weights whose second moment is
    # far below 2^-96, single tiny elements — take the out-of-line replay by the compiler's expansions: touched rows
    # (catch-up + sparse step) and rows no batch ever touches (flush) alike
    U[7] = 0.0
    U[11] = 1e-30
    U[13, ::3] = 0.0
    A[5] = 0.0
    A[6, 17] = 3e-33
    never = np.setdiff1d(np.arange(n_u), np.unique(ui))[:3]
    assert len(never) == 3
    U[never[0]] = 0.0
    U[never[1]] = -2e-31
    U[never[2], 5] = 0.0
    n = len(ui)
    starts, counts, alphas = _schedule(n, B, 2e-4)
    engs = {}
    for lazy in (False, True):
        eng = TrainEngine(n_u, n_a, max_batch=B, arena_steps=16, lazy=lazy)
        assert eng.lazy == lazy
        eng.set_head(w=1.2)
        eng.set_weights(U, A)
        eng.set_epoch(ui, ai, t, starts, counts, alphas)
        done = 0
        for chunk in (1, 3, 8, 21, 9, 35):          # windows cut short, exactly full, several per call
            done += eng.run(chunk, use_graph=use_graph, first_step=done)
        assert done == len(counts) == steps
        eng.synchronize()            # the engine runs on its own stream: finish before torch reads the tables
        engs[lazy] = eng
    d, z = engs[False], engs[True]
    for name, x, y in (("W", d.W, z.W), ("M", d.M, z.M), ("V", d.V, z.V)):
        if not torch.equal(x, y):
            bad = torch.nonzero((x != y).any(1)).flatten()
            raise AssertionError("%s differs in %d rows (first %s), max |diff| %.3e" % (
                name, bad.numel(), bad[:8].tolist(), float((x - y).abs().max())))
    rd, rz = d.read_state(), z.read_state()
    for k in ("w", "b", "gamma", "beta", "adam_m", "adam_v", "mov_mean", "mov_var", "bn_mu", "bn_var", "last_mse",
              "se_sum", "n_seen", "bce_wsum", "step_fwd"):
        assert np.array_equal(rd[k], rz[k]), k
    for k in ("last_loss", "reg_sumsq", "reg_user_sumsq", "reg_anime_sumsq"):
        assert abs(float(rd[k]) - float(rz[k])) <= 2e-6 * abs(float(rd[k])) + 1e-7, k
    for k in ("loss_wsum", "reg_user_wsum", "reg_anime_wsum"):
        assert abs(float(rd[k]) - float(rz[k])) <= 2e-6 * abs(float(rd[k])), k
    # a dense stage-by-stage step after lazy runs sees consistent L2 partials (the flush leaves them)
    for eng in (d, z):
        eng.set_epoch(ui, ai, t, starts, counts, alphas)
        eng.run(5, use_graph=False)
        eng.prep(5, 1)
        eng.fwd(); eng.head(); eng.bwd(); eng.adam()
        eng.synchronize()
    rd, rz = d.read_state(), z.read_state()
    assert torch.equal(d.W, z.W) and abs(float(rd["last_loss"]) - float(rz["last_loss"])) <= 3e-6 * abs(float(rd["last_loss"]))
    d.close(); z.close()


def test_fit_with_the_lazy_update_returns_the_dense_history_and_tables():
    """`trainer.fit` end to end (epoch shuffles, ragged last batch, graph blocks + eager tail, validation between the
    epochs, best-weights snapshot) over an engine with the lazy dense Adam and one with the dense kernel: the tables,
    the head and every optimizer slot `model.save` would hold are bit-identical; the History columns agree to the
    rounding of the L2 sum inside the loss (mse, val_mse and lr are equal)."""
    from anime_recommendations_amd import data, trainer
    from anime_recommendations_amd.engine import TrainEngine
    df = data.synth_user_stats(n_users=6000, n_anime=700, n_ratings=40_000, seed=11)
    table = data.encode_frame(df)
    cfg = trainer.FitConfig(epochs=4, batch_size=700, test_size=1500, verbose=0, seed=3, arena_steps=16)
    res = {}
    for lazy in (False, True):
        eng = TrainEngine(table.n_users, table.n_anime, max_batch=cfg.batch_size, l2=cfg.l2_reg_factor,
                          arena_steps=cfg.arena_steps, lazy=lazy)
        assert eng.lazy == lazy
        res[lazy] = trainer.fit(table, cfg, engine=eng)
        eng.close()
    d, z = res[False], res[True]
    assert np.array_equal(d.U, z.U) and np.array_equal(d.A, z.A) and d.head == z.head
    assert np.array_equal(d.best_U, z.best_U) and np.array_equal(d.best_A, z.best_A) and d.best_epoch == z.best_epoch
    assert sorted(d.optimizer) == sorted(z.optimizer)
    for k in d.optimizer:
        assert np.array_equal(np.asarray(d.optimizer[k]), np.asarray(z.optimizer[k])), k
    for k in ("mse", "val_mse", "lr"):
        assert d.history[k] == z.history[k], k
    for k in ("loss", "val_loss"):
        np.testing.assert_allclose(d.history[k], z.history[k], rtol=3e-6, atol=0)
