"""The plain-C oracle (cpu_baseline port + exact fmaf-chain checker) against the NumPy oracle."""
import numpy as np

from oracle import anirec_oracle as orc
from oracle import c_oracle


def _problem(seed, n_u, n_a, n):
    rng = np.random.default_rng(seed)
    U = rng.uniform(-0.05, 0.05, (n_u, 128)).astype(np.float32)
    A = rng.uniform(-0.05, 0.05, (n_a, 128)).astype(np.float32)
    ui = rng.integers(0, n_u, n)
    ai = (rng.zipf(1.2, n) - 1) % n_a
    t = (rng.integers(0, 11, n) / 10).astype(np.float32)
    return U, A, ui, ai, t


def test_c_train_steps_match_numpy_oracle():
    U, A, ui, ai, t = _problem(0, 500, 300, 5 * 400 - 100)
    B, lr = 400, 3e-5
    s_np = orc.new_state(U, A, orc.new_head(w=1.2))
    s_c = orc.new_state(U, A, orc.new_head(w=1.2))
    alphas = []
    last = None
    for k, st in enumerate(range(0, len(ui), B)):
        last, _, _ = orc.train_step(s_np, ui[st:st + B], ai[st:st + B], t[st:st + B], lr)
        alphas.append(orc.adam_alpha(lr, k + 1))
    met = c_oracle.train_run(s_c, ui, ai, t, B, alphas)
    tol = lr * 2e-3 * len(alphas)
    np.testing.assert_allclose(s_c["U"], s_np["U"], atol=tol)
    np.testing.assert_allclose(s_c["A"], s_np["A"], atol=tol)
    np.testing.assert_allclose(s_c["vA"], s_np["vA"], atol=np.abs(s_np["vA"]).max() * 1e-4)
    assert abs(met["loss"] - float(last["loss"])) < 3e-6
    assert abs(met["mse"] - float(last["mse"])) < 1e-6
    for k in ("w", "gamma", "beta", "mov_mean", "mov_var"):
        assert abs(float(s_c["head"][k]) - float(s_np["head"][k])) < max(tol, 1e-6), k


def test_c_fmaf_chain_vs_numpy_emulation_and_topk():
    rng = np.random.default_rng(1)
    W = rng.normal(0, 0.05, (400, 128)).astype(np.float32)
    W[9] = W[2]
    Wh = orc.rownorm(W)
    s_c = c_oracle.cosine_scores(Wh, 7)
    s_np = orc.dot_chain_f32(Wh, Wh[7])
    assert (s_c == s_np).mean() > 0.999
    np.testing.assert_allclose(s_c, s_np, atol=1.2e-7)
    # top-k of the C oracle == NumPy top-k on the C scores (tie rule: ascending index)
    oi, ov = c_oracle.cosine_topk(Wh, [2, 7, 399], 20)
    for j, q in enumerate([2, 7, 399]):
        ii, ss = orc.topk_desc(c_oracle.cosine_scores(Wh, q), 20, exclude=q)
        assert (oi[j] == ii).all() and (ov[j] == ss).all()
    assert oi[0, 0] == 9
    keep = np.zeros(400, np.uint8)
    keep[[1, 5, 9]] = 1
    oi, ov = c_oracle.cosine_topk(Wh, [2], 5, keep=keep)
    assert set(oi[0, :3]) == {1, 5, 9} and (oi[0, 3:] == -1).all() and np.isnan(ov[0, 3:]).all()


def test_c_rownorm_and_predict_grid():
    rng = np.random.default_rng(2)
    U = rng.normal(0, 0.05, (30, 128)).astype(np.float32)
    A = rng.normal(0, 0.05, (50, 128)).astype(np.float32)
    np.testing.assert_allclose(c_oracle.rownorm(A), orc.rownorm(A), rtol=5e-7)  # 1-2 ulp: sum order
    head = orc.new_head(w=1.3, b=0.1, gamma=0.9, beta=-0.2, mov_mean=0.05, mov_var=0.4)
    np.testing.assert_allclose(c_oracle.predict_grid(U, A, head, [3, 29]),
                               orc.predict_grid(U, A, head, [3, 29]), atol=2e-7)
