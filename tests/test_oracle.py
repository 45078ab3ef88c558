"""Oracle self-checks (CPU).  The oracle is the parity spec; these pin it to the
only numeric known-answer the reference holds (the lr column of its history CSV)
and cross-check the closed-form backward against torch autograd in fp64."""
import csv
import os

import numpy as np
import pytest
import torch

from oracle import anirec_oracle as orc


def test_lrfn_matches_reference_history_csv(golden_dir):
    # reference: figure_file/anime_nn_history.csv (lr column), produced by
    # neural_network.py:109-125 with config.yaml:57-62 (+ start_lr 1e-5)
    with open(os.path.join(golden_dir, "anime_nn_history.csv")) as f:
        rows = list(csv.DictReader(f))
    assert list(rows[0].keys()) == ["", "loss", "mse", "val_loss", "val_mse", "lr"]
    assert len(rows) == 15
    for e, r in enumerate(rows):
        got = np.float32(orc.lrfn(e))
        want = np.float32(float(r["lr"]))
        assert got == want, (e, got, want)


def _torch_loss(U, A, w, b, gamma, beta, ui, ai, t, l2):
    u = U[ui]
    a = A[ai]
    uh = u * torch.rsqrt(torch.clamp((u * u).sum(1, keepdim=True), min=orc.L2N_EPS))
    ah = a * torch.rsqrt(torch.clamp((a * a).sum(1, keepdim=True), min=orc.L2N_EPS))
    c = (uh * ah).sum(1)
    z = c * w + b
    mu = z.mean()
    var = ((z - mu.detach()) ** 2).mean()
    y = (z - mu) * torch.rsqrt(var + orc.BN_EPS) * gamma + beta
    bce = torch.nn.functional.binary_cross_entropy_with_logits(y, t)
    return bce + l2 * ((U * U).sum() + (A * A).sum())


def test_closed_form_backward_matches_autograd_fp64():
    rng = np.random.default_rng(0)
    n_u, n_a, D, B = 40, 30, 128, 64
    U = rng.uniform(-0.05, 0.05, (n_u, D))
    A = rng.uniform(-0.05, 0.05, (n_a, D))
    ui = rng.integers(0, n_u, B)
    ai = rng.integers(0, n_a, B)
    ai[:8] = 3                       # duplicates
    t = rng.integers(0, 11, B) / 10.0
    head = orc.new_head(w=1.3, b=0.1, gamma=0.9, beta=-0.2)
    f, g, met = orc.grads(U, A, ui, ai, t, head, l2=1e-4, dtype=np.float64)

    tU = torch.tensor(U, requires_grad=True)
    tA = torch.tensor(A, requires_grad=True)
    sc = [torch.tensor(float(head[k]), dtype=torch.float64, requires_grad=True)
          for k in ("w", "b", "gamma", "beta")]
    loss = _torch_loss(tU, tA, *sc, torch.tensor(ui), torch.tensor(ai), torch.tensor(t), 1e-4)
    loss.backward()
    assert abs(float(loss.detach()) - float(met["loss"])) < 1e-12
    np.testing.assert_allclose(g["U"], tU.grad.numpy(), rtol=0, atol=1e-14)
    np.testing.assert_allclose(g["A"], tA.grad.numpy(), rtol=0, atol=1e-14)
    for k, s in zip(("w", "b", "gamma", "beta"), sc):
        assert abs(float(g[k]) - float(s.grad)) < 1e-13, k


def test_fp32_step_close_to_fp64_step():
    rng = np.random.default_rng(1)
    n_u, n_a, D, B = 64, 48, 128, 32
    U = rng.uniform(-0.05, 0.05, (n_u, D)).astype(np.float32)
    A = rng.uniform(-0.05, 0.05, (n_a, D)).astype(np.float32)
    ui = rng.integers(0, n_u, B)
    ai = rng.integers(0, n_a, B)
    t = (rng.integers(0, 11, B) / 10.0).astype(np.float32)
    s32 = orc.new_state(U, A, orc.new_head(w=1.1))
    s64 = orc.new_state(U.astype(np.float64), A.astype(np.float64), orc.new_head(w=1.1))
    for k in ("mU", "vU", "mA", "vA"):
        s64[k] = s64[k].astype(np.float64)
    for _ in range(3):
        m32, _, _ = orc.train_step(s32, ui, ai, t, lr=1e-5, dtype=np.float32)
        m64, _, _ = orc.train_step(s64, ui, ai, t, lr=1e-5, dtype=np.float64)
    assert abs(float(m32["loss"]) - float(m64["loss"])) < 1e-5
    np.testing.assert_allclose(s32["U"], s64["U"], atol=2e-6)
    np.testing.assert_allclose(s32["A"], s64["A"], atol=2e-6)


def test_adam_first_step_is_sign_step():
    W = np.full((2, 4), 0.5, np.float32)
    m = np.zeros_like(W)
    v = np.zeros_like(W)
    g = np.array([[1.0, -1.0, 20.0, -5.0]] * 2, np.float32)
    a = orc.adam_alpha(1e-5, 1)
    orc.adam_update(W, m, v, g, a)
    # t=1: alpha = lr*sqrt(1-b2)/(1-b1), m/sqrt(v) = 0.1*g/(sqrt(0.001)*|g|)
    np.testing.assert_allclose(W, 0.5 - np.sign(g) * 1e-5, atol=2e-9)


def test_zero_row_forward_and_grad_are_finite():
    U = np.zeros((2, 128), np.float32)
    A = np.random.default_rng(2).uniform(-0.05, 0.05, (3, 128)).astype(np.float32)
    f, g, met = orc.grads(U, A, np.array([0, 1]), np.array([1, 2]), np.array([0.5, 1.0], np.float32),
                          orc.new_head())
    assert np.all(f["c"] == 0)
    assert np.isfinite(g["U"]).all() and np.isfinite(g["A"]).all()


def test_rownorm_and_topk_tie_rule():
    W = np.random.default_rng(3).normal(size=(50, 128)).astype(np.float32)
    W[7] = W[3]                      # exact tie with row 3 for every query
    Wh = orc.rownorm(W)
    assert Wh.dtype == np.float32
    np.testing.assert_allclose(np.linalg.norm(Wh, axis=1), 1.0, atol=1e-6)
    idx, sim = orc.cosine_topk(Wh, [3, 10], 5)
    assert idx[0, 0] == 7 and 3 not in idx[0]
    i10 = list(idx[1])
    if 3 in i10 and 7 in i10:
        assert i10.index(3) + 1 == i10.index(7)   # ascending index among equals
    assert np.all(np.diff(sim, axis=1) <= 0)
    # zero row -> NaN (no epsilon in the reference) and is ranked last
    W[5] = 0
    with np.errstate(invalid="ignore", divide="ignore"):
        Wh = orc.rownorm(W)
    assert np.isnan(Wh[5]).all()
    idx, _ = orc.cosine_topk(Wh, [0], 49)
    assert idx[0, -1] == 5


def test_predict_matches_eval_forward():
    rng = np.random.default_rng(4)
    U = rng.normal(0, 0.05, (9, 128)).astype(np.float32)
    A = rng.normal(0, 0.05, (11, 128)).astype(np.float32)
    head = orc.new_head(w=1.3, b=0.1, gamma=0.9, beta=-0.2, mov_mean=0.05, mov_var=0.4)
    G = orc.predict_grid(U, A, head, [2, 5])
    assert G.shape == (2, 11) and np.all((G > 0) & (G < 1))
    p = orc.predict_pairs(U, A, head, np.array([5]), np.array([7]))
    assert p[0] == G[1, 7]


def test_ingest_oracle_hand_example():
    """Pins the pandas restatement of preprocess.py on a frame small enough to do by hand."""
    import pandas as pd
    from oracle import ingest_oracle as ing
    df = pd.DataFrame({
        "user_id":          [10, 10, 20, 10, 20, 30, 10, 20],
        "anime_id":         [5,  6,  5,  5,  7,  5,  8,  9],
        "rating":           [8,  2,  10, 8,  6,  9,  4,  np.nan],
        "watching_status":  [2,  2,  2,  2,  6,  2,  2,  2],
        "watched_episodes": [12, 0,  12, 12, 3,  1,  6,  2],
    })
    # row 3 duplicates row 0; row 7 has a NaN; user 30 has one rating
    out = ing.preprocess(df, num_reviews=2)
    assert out["user_id"].tolist() == [10, 10, 20, 20, 10]
    assert out["anime_id"].tolist() == [5, 6, 5, 7, 8]
    assert out["rating"].tolist() == [(8 - 2) / 8, 0.0, 1.0, (6 - 2) / 8, (4 - 2) / 8]
    out = ing.preprocess(df, num_reviews=2, drop_unwatched=True, drop_plan=True)
    # after the row filters user 20 keeps one rating and is dropped; user 10 keeps rows 0 and 6
    assert out["user_id"].tolist() == [10, 10] and out["rating"].tolist() == [1.0, 0.0]
    out = ing.preprocess(df, num_reviews=1, drop_half=True)
    # per-anime max watched: 5 -> 12, 6 -> 0, 7 -> 3, 8 -> 6; user 30's single episode of anime 5 is < 6
    assert out["anime_id"].tolist() == [5, 6, 5, 7, 8]
    idx, uniq = ing.encode(pd.Series([7, 3, 7, 9, 3]))
    assert idx.tolist() == [0, 1, 0, 2, 1] and uniq.tolist() == [7, 3, 9]


def test_recs_oracle_hand_example():
    """Pins the NumPy/pandas restatement of fave_genres / similar_user_recs on a case done by hand."""
    from oracle import recs_oracle as rec
    # user 0 rates five anime 0.2 .. 1.0: 80th percentile of [.2,.4,.6,.8,1.] = .2 + .8*4*(.2) -> position 3.2
    u = np.array([0, 0, 0, 0, 0, 1, 1, 2, 2, 2])
    a = np.array([0, 1, 2, 3, 4, 3, 4, 4, 5, 6])
    r = np.array([.2, .4, .6, .8, 1., .5, .7, .9, .9, .1])
    thr, fav = rec.favourites(u, a, r, 4)
    assert thr[0] == np.percentile([.2, .4, .6, .8, 1.], 80) and abs(thr[0] - 0.84) < 1e-12
    assert fav[0] == {4} and fav[1] == {4} and fav[2] == {4, 5} and fav[3] == set() and np.isnan(thr[3])
    # query user 3 (no favourites of its own), similar users 2, 0, 1 (best first): anime 4 three times, 5 once
    order, counts = rec.user_recs(fav, 3, [2, 0, 1], 5)
    assert order == [4, 5] and counts == [3, 1]
    assert rec.value_counts_of_similar_favourites(fav, 3, [2, 0, 1]) == {4: 3, 5: 1}
    # query user 0 holds anime 4 itself: only 5 is left
    assert rec.user_recs(fav, 0, [2, 1], 5) == ([5], [1])
