"""CPU oracle for the anime_recommendations hot path.

TEST INFRASTRUCTURE ONLY.  Nothing outside ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product
(``anime_recommendations_amd``) never routes through it and has no CPU fallback.

PARITY PINNING: the reference has no tests and TensorFlow/Keras 2.12 (where the
arithmetic lives, un-vendored: ``neural_network/conda.yml:20-21``) is not
installed here, so this restates the *published* Keras 2.12 / NumPy semantics
at the reference's call sites.  What IS pinned to the reference itself
(tests/test_reference_fixtures.py, fixtures generated from the reference's own
function bodies by tests/golden/make_reference_function_fixtures.py, under
numpy 2.2 / pandas 2.3 — not the reference's pinned 1.23.5 / 1.5.3):
``lrfn`` (also the ``lr`` column of ``figure_file/anime_nn_history.csv``),
``rownorm`` (= get_weights), the cosine neighbour lists of ``find_similar_users``,
and — in ingest_oracle.py / recs_oracle.py — preprocess, favourites, ``clean`` /
``by_genre``, ``get_unwatched``.  What stays **parity unpinned**: everything
that executes inside Keras/TensorFlow — the forward graph, the train step
(closed-form backward + Adam), validation metrics and ``model.predict`` —
because nothing in the reference holds a vector for it.

Each function cites the reference file:line it follows.  ``dtype`` selects the
arithmetic type: ``np.float32`` is the parity oracle (what Keras computes in),
``np.float64`` is used to sanity-check the closed-form backward.
"""
from __future__ import annotations

import numpy as np

# ---------------------------------------------------------------------------
# constants (Keras 2.12 defaults at the reference call sites)
# ---------------------------------------------------------------------------
L2N_EPS = 1e-12      # tf.nn.l2_normalize epsilon used by Dot(normalize=True)   neural_network.py:90-93
BN_EPS = 1e-3        # BatchNormalization() default epsilon                      neural_network.py:99
BN_MOMENTUM = 0.99   # BatchNormalization() default momentum                     neural_network.py:99
ADAM_B1 = 0.9        # optimizer='Adam' defaults                                 neural_network.py:104
ADAM_B2 = 0.999
ADAM_EPS = 1e-7


# ---------------------------------------------------------------------------
# a4: learning-rate schedule                                  neural_network.py:109-125
# ---------------------------------------------------------------------------
def lrfn(epoch, start_lr=1e-5, max_lr=5e-5, min_lr=1e-5, rampup_epochs=5,
         sustain_epochs=0, exp_decay=0.8):
    """Per-epoch learning rate, python floats exactly as the reference computes it."""
    max_lr = float(max_lr)
    if epoch < int(rampup_epochs):
        return (float(max_lr) - float(start_lr)) / int(rampup_epochs) * epoch + float(start_lr)
    elif epoch < int(rampup_epochs) + int(sustain_epochs):
        return float(max_lr)
    else:
        return (float(max_lr) - float(min_lr)) * float(exp_decay) ** (
            epoch - int(rampup_epochs) - int(sustain_epochs)) + float(min_lr)


def adam_alpha(lr, t, dtype=np.float32):
    """Keras-2.12 Adam bias-corrected step size for iteration ``t`` (1-based).

    optimizer='Adam' (neural_network.py:104) -> keras.optimizers.Adam.update_step:
    ``alpha = lr * sqrt(1 - beta_2**t) / (1 - beta_1**t)`` evaluated in the
    variable dtype.
    """
    dt = dtype
    lr = dt(lr)
    ts = dt(t)
    b1p = np.power(dt(ADAM_B1), ts, dtype=dt)
    b2p = np.power(dt(ADAM_B2), ts, dtype=dt)
    return dt(lr * np.sqrt(dt(1) - b2p, dtype=dt) / (dt(1) - b1p))


# ---------------------------------------------------------------------------
# a1: index contract                                           neural_network.py:41-60
# ---------------------------------------------------------------------------
def encode_ids(ids):
    """id -> dense index in order of first appearance (``Series.unique()`` order).

    Returns (index_array, uniques).  neural_network.py:43-52.
    """
    ids = np.asarray(ids)
    uniq, first = np.unique(ids, return_index=True)
    order = np.argsort(first, kind="stable")
    uniques = uniq[order]
    rank_of_sorted = np.empty(len(uniq), dtype=np.int64)
    rank_of_sorted[order] = np.arange(len(uniq))
    idx = rank_of_sorted[np.searchsorted(uniq, ids)]
    return idx, uniques


def shuffle_rows(n, random_state=42):
    """Row order of ``df.sample(frac=1, random_state=42)``  (neural_network.py:59).

    pandas' sample(frac=1) draws ``RandomState(seed).permutation(n)``-equivalent
    row positions (checked here on pandas 2.3.3 by tests/test_data.py).
    """
    return np.random.RandomState(random_state).permutation(n)


# ---------------------------------------------------------------------------
# a2: forward graph                                            neural_network.py:66-106
# ---------------------------------------------------------------------------
def _inv_norm(ss, dt):
    """rsqrt(max(sum_sq, eps)) as 1/sqrt in the working dtype (tf.nn.l2_normalize)."""
    return dt(1) / np.sqrt(np.maximum(ss, dt(L2N_EPS)), dtype=dt)


def forward(U, A, ui, ai, head, training, dtype=np.float32):
    """Embedding x2 -> Dot(normalize=True) -> Dense(1) -> BatchNorm -> sigmoid.

    head: dict with w, b, gamma, beta, mov_mean, mov_var (scalars).
    Returns dict of every intermediate (all ``dtype``).
    """
    dt = dtype
    u = U[ui].astype(dt)
    a = A[ai].astype(dt)
    su = np.sum(u * u, axis=1, dtype=dt)
    sa = np.sum(a * a, axis=1, dtype=dt)
    ru = _inv_norm(su, dt)
    ra = _inv_norm(sa, dt)
    uh = u * ru[:, None]
    ah = a * ra[:, None]
    c = np.sum(uh * ah, axis=1, dtype=dt)                      # Dot(normalize=True, axes=2)
    w, b = dt(head["w"]), dt(head["b"])
    z = c * w + b                                              # Dense(1)
    gamma, beta = dt(head["gamma"]), dt(head["beta"])
    if training:                                               # BatchNormalization, non-fused 2-D path
        mu = np.mean(z, dtype=dt)
        var = np.mean((z - mu) * (z - mu), dtype=dt)           # tf.nn.moments: biased
    else:
        mu, var = dt(head["mov_mean"]), dt(head["mov_var"])
    r = dt(1) / np.sqrt(var + dt(BN_EPS), dtype=dt)            # rsqrt(var + eps)
    inv = r * gamma                                            # tf.nn.batch_normalization
    y = z * inv + (beta - mu * inv)
    p = _sigmoid(y, dt)                                        # Activation('sigmoid')
    return dict(u=u, a=a, su=su, sa=sa, ru=ru, ra=ra, uh=uh, ah=ah, c=c, z=z,
                mu=mu, var=var, r=r, y=y, p=p)


def _sigmoid(y, dt):
    y = np.asarray(y, dtype=dt)
    e = np.exp(-np.abs(y), dtype=dt)
    return np.where(y >= 0, dt(1) / (dt(1) + e), e / (dt(1) + e)).astype(dt)


def bce_from_logits(y, t, dt=np.float32):
    """tf.nn.sigmoid_cross_entropy_with_logits: max(y,0) - y*t + log1p(exp(-|y|)).

    Keras 2.12 ``binary_crossentropy`` takes this branch because the sigmoid
    activation caches ``_keras_logits`` (neural_network.py:100-104).
    """
    y = np.asarray(y, dtype=dt)
    t = np.asarray(t, dtype=dt)
    return np.maximum(y, dt(0)) - y * t + np.log1p(np.exp(-np.abs(y), dtype=dt), dtype=dt)


def reg_sumsq(U, A, dtype=np.float32):
    """sum(U^2) + sum(A^2): the embeddings_regularizer L2 term without lambda
    (neural_network.py:73,78,85)."""
    dt = dtype
    return dt(np.sum(np.square(U.astype(dt)), dtype=dt) + np.sum(np.square(A.astype(dt)), dtype=dt))


# ---------------------------------------------------------------------------
# a3: one training step (forward, loss, closed-form backward, dense Adam)
#                                                              neural_network.py:210-217
# ---------------------------------------------------------------------------
def new_head(w=1.0, b=0.0, gamma=1.0, beta=0.0, mov_mean=0.0, mov_var=1.0):
    """Scalar head state at Keras init: Dense bias 0, BN gamma 1 / beta 0,
    moving mean 0 / variance 1; ``w`` is the he_normal draw (neural_network.py:97-99)."""
    return dict(w=np.float32(w), b=np.float32(b), gamma=np.float32(gamma), beta=np.float32(beta),
                mov_mean=np.float32(mov_mean), mov_var=np.float32(mov_var),
                m=np.zeros(4, np.float32), v=np.zeros(4, np.float32))


def new_state(U, A, head=None):
    """Full trainable + optimizer state (tables, Adam slots, head, iteration count)."""
    return dict(U=U.copy(), A=A.copy(),
                mU=np.zeros_like(U), vU=np.zeros_like(U),
                mA=np.zeros_like(A), vA=np.zeros_like(A),
                head=head if head is not None else new_head(), t=0)


def grads(U, A, ui, ai, t, head, l2=1e-4, dtype=np.float32):
    """Loss and every gradient of one batch (training-mode BN).

    Returns (fwd dict, grads dict, metrics dict).  gU/gA are the DENSE gradients
    the Keras step hands to Adam: scatter-added gather gradients (IndexedSlices,
    duplicates summed in batch order) plus the dense regulariser term 2*l2*W.
    """
    dt = dtype
    f = forward(U, A, ui, ai, head, training=True, dtype=dt)
    tt = np.asarray(t, dtype=dt)
    B = dt(len(tt))
    w, gamma = dt(head["w"]), dt(head["gamma"])
    li = bce_from_logits(f["y"], tt, dt)
    bce = np.sum(li, dtype=dt) / B
    reg = reg_sumsq(U, A, dt)
    loss = bce + dt(l2) * reg
    mse = np.sum((f["p"] - tt) ** 2, dtype=dt) / B            # metrics=['mse']  config.yaml:88

    dy = (f["p"] - tt) / B
    zhat = (f["z"] - f["mu"]) * f["r"]
    d_beta = np.sum(dy, dtype=dt)
    d_gamma = np.sum(dy * zhat, dtype=dt)
    dzh = dy * gamma
    m1 = np.sum(dzh, dtype=dt) / B
    m2 = np.sum(dzh * zhat, dtype=dt) / B
    dz = (dzh - m1 - zhat * m2) * f["r"]
    d_w = np.sum(dz * f["c"], dtype=dt)
    d_b = np.sum(dz, dtype=dt)
    dc = dz * w
    # d c / d u = ru * (ah - c*uh) when sum(u^2) >= eps, else ru * ah (the max() gates the norm path)
    coef = dc * f["ru"] * f["ra"]
    self_u = np.where(f["su"] >= dt(L2N_EPS), dc * f["c"] * f["ru"] * f["ru"], dt(0)).astype(dt)
    self_a = np.where(f["sa"] >= dt(L2N_EPS), dc * f["c"] * f["ra"] * f["ra"], dt(0)).astype(dt)
    du = coef[:, None] * f["a"] - self_u[:, None] * f["u"]
    da = coef[:, None] * f["u"] - self_a[:, None] * f["a"]
    gU = np.zeros(U.shape, dt)
    gA = np.zeros(A.shape, dt)
    np.add.at(gU, ui, du)
    np.add.at(gA, ai, da)
    two_l2 = dt(2.0 * l2)
    gU = gU + two_l2 * U.astype(dt)
    gA = gA + two_l2 * A.astype(dt)
    g = dict(U=gU, A=gA, w=d_w, b=d_b, gamma=d_gamma, beta=d_beta,
             dc=dc, coef=coef, self_u=self_u, self_a=self_a)
    met = dict(loss=loss, bce=bce, reg=reg, mse=mse)
    return f, g, met


def adam_update(W, m, v, g, alpha, dtype=np.float32):
    """Keras-2.12 Adam dense branch, in place.

    m += (g - m)*(1-b1);  v += (g*g - v)*(1-b2);  W -= (m*alpha)/(sqrt(v)+eps)
    """
    dt = dtype
    one_b1 = dt(1.0 - ADAM_B1)
    one_b2 = dt(1.0 - ADAM_B2)
    m += (g - m) * one_b1
    v += (g * g - v) * one_b2
    W -= (m * dt(alpha)) / (np.sqrt(v, dtype=dt) + dt(ADAM_EPS))


def train_step(state, ui, ai, t, lr, l2=1e-4, dtype=np.float32):
    """One ``model.fit`` step on one batch; mutates ``state``; returns metrics."""
    dt = dtype
    head = state["head"]
    f, g, met = grads(state["U"], state["A"], ui, ai, t, head, l2, dt)
    state["t"] += 1
    alpha = adam_alpha(lr, state["t"], dt)
    adam_update(state["U"], state["mU"], state["vU"], g["U"], alpha, dt)
    adam_update(state["A"], state["mA"], state["vA"], g["A"], alpha, dt)
    hp = np.array([head["w"], head["b"], head["gamma"], head["beta"]], dt)
    hg = np.array([g["w"], g["b"], g["gamma"], g["beta"]], dt)
    hm = head["m"].astype(dt)
    hv = head["v"].astype(dt)
    adam_update(hp, hm, hv, hg, alpha, dt)
    head["w"], head["b"], head["gamma"], head["beta"] = hp
    head["m"], head["v"] = hm, hv
    # moving stats: variable -= (variable - batch) * (1 - momentum); biased batch variance
    dec = dt(1.0 - BN_MOMENTUM)
    head["mov_mean"] = dt(head["mov_mean"]) - (dt(head["mov_mean"]) - f["mu"]) * dec
    head["mov_var"] = dt(head["mov_var"]) - (dt(head["mov_var"]) - f["var"]) * dec
    met = dict(met)
    met.update(alpha=alpha, mu=f["mu"], var=f["var"])
    return met, f, g


def evaluate(state, ui, ai, t, l2=1e-4, dtype=np.float32):
    """Validation pass (BN inference mode); val_loss includes the L2 term
    (neural_network.py:216)."""
    dt = dtype
    f = forward(state["U"], state["A"], ui, ai, state["head"], training=False, dtype=dt)
    tt = np.asarray(t, dt)
    B = dt(len(tt))
    bce = np.sum(bce_from_logits(f["y"], tt, dt), dtype=dt) / B
    loss = bce + dt(l2) * reg_sumsq(state["U"], state["A"], dt)
    mse = np.sum((f["p"] - tt) ** 2, dtype=dt) / B
    return dict(val_loss=loss, val_mse=mse, bce=bce, p=f["p"])


# ---------------------------------------------------------------------------
# a6: get_weights row normalisation            similar_anime.py:136-171 (and 3 copies)
# ---------------------------------------------------------------------------
def rownorm(W):
    """``W / np.linalg.norm(W, axis=1).reshape(-1, 1)`` — fp32 in, fp32 out, no epsilon."""
    W = np.asarray(W)
    return W / np.linalg.norm(W, axis=1).reshape((-1, 1))


# ---------------------------------------------------------------------------
# a7/a8: cosine of query rows vs all rows + top-k
#        similar_anime.py:404-408 ; similar_users.py:293-296
# ---------------------------------------------------------------------------
def dot_chain_f32(Wh, q):
    """fp32 dot products with the FIXED summation order the build defines:
    a k-ordered fused-multiply-add chain  s = fma(w[k], q[k], s), k = 0..D-1
    (the order of v_mfma_f32_32x32x2_f32).  Emulated exactly: fp32 products are
    exact in fp64 (24+24 <= 53 bits) and the fp64 sum of an exact product and an
    fp32 value, rounded to fp32, is correctly rounded except for double-rounding
    ties which the C oracle (fmaf) removes; tests use the C oracle for bit checks.
    """
    Wh = np.asarray(Wh, np.float32)
    q = np.asarray(q, np.float32)
    s = np.zeros(Wh.shape[0], np.float32)
    for k in range(Wh.shape[1]):
        s = (Wh[:, k].astype(np.float64) * np.float64(q[k]) + s.astype(np.float64)).astype(np.float32)
    return s


def topk_desc(scores, k, exclude=None, mask=None):
    """Top-k by descending score; tie rule (the build's definition, the reference's
    ``np.argsort`` is an unstable introsort so ties are undefined there): equal
    scores order by ascending index.  NaN scores (zero rows) sort last.
    ``exclude``: index dropped (the query itself, similar_users.py:303 /
    similar_anime.py:459).  ``mask``: optional boolean keep-mask (Type/Genre filters)."""
    s = np.asarray(scores, np.float32).copy()
    n = len(s)
    keep = np.ones(n, bool) if mask is None else np.asarray(mask, bool).copy()
    if exclude is not None and 0 <= exclude < n:
        keep[exclude] = False
    key = np.where(np.isnan(s), -np.inf, s)
    idx = np.nonzero(keep)[0]
    order = np.lexsort((idx, -key[idx].astype(np.float64)))
    sel = idx[order][:k]
    return sel.astype(np.int64), s[sel]


def cosine_topk(Wh, queries, k, exclude_self=True, mask=None):
    """For each query row index: scores = Wh @ Wh[q] (fixed-order fp32), top-k desc."""
    out_i = np.full((len(queries), k), -1, np.int64)
    out_s = np.full((len(queries), k), np.nan, np.float32)
    for j, q in enumerate(queries):
        s = dot_chain_f32(Wh, Wh[q])
        ii, ss = topk_desc(s, k, exclude=q if exclude_self else None, mask=mask)
        out_i[j, :len(ii)] = ii
        out_s[j, :len(ss)] = ss
    return out_i, out_s


# ---------------------------------------------------------------------------
# a9: model.predict on (user, anime) pairs          model_recs.py:394 (+ :396 ranking)
# ---------------------------------------------------------------------------
def predict_pairs(U, A, head, ui, ai, dtype=np.float32):
    """``model.predict([user_arr, anime_arr]).flatten()`` — BN in inference mode."""
    return forward(U, A, ui, ai, head, training=False, dtype=dtype)["p"]


def predict_grid(U, A, head, users, dtype=np.float32):
    """Predicted rating of every anime for each user in ``users`` -> (len(users), n_anime)."""
    n_a = A.shape[0]
    out = np.empty((len(users), n_a), dtype)
    ai = np.arange(n_a)
    for j, u in enumerate(users):
        out[j] = predict_pairs(U, A, head, np.full(n_a, u), ai, dtype)
    return out
