"""Soak of the lazy dense Adam against the dense kernel (bitwise) over random shapes, batch sizes, learning rates, run()
chunkings, graph / eager, arena sizes and ragged last batches: python scripts/soak_lazy_adam.py [n_configs] [seed0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import anirec_oracle as orc
from anime_recommendations_amd.engine import TrainEngine
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t0 = time.time()
for c in range(n_cfg):
    rng = np.random.default_rng(7000 + seed0 + c)
    n_u = int(rng.integers(1500, 60000)); n_a = int(rng.integers(200, 5000))
    B = int(rng.choice([64, 200, 512, 1000, 2048, 4096]))
    steps = int(rng.integers(20, 110))
    n = B * steps - int(rng.integers(0, B))            # ragged last batch
    U = rng.uniform(-0.05, 0.05, (n_u, 128)).astype(np.float32)
    A = rng.uniform(-0.05, 0.05, (n_a, 128)).astype(np.float32)
    if rng.random() < 0.5:                             # rows off the replay's short arithmetic range
        U[rng.integers(0, n_u, 3)] = 0.0
        A[rng.integers(0, n_a)] = 1e-30
        U[rng.integers(0, n_u), ::5] = 0.0
    ui = rng.integers(0, n_u, n).astype(np.int64)
    ai = ((rng.zipf(1.0 + rng.random() * 0.4 + 0.05, n) - 1) % n_a).astype(np.int64)
    t = (rng.integers(0, 11, n) / 10).astype(np.float32)
    lr = float(10 ** rng.uniform(-5.5, -3))
    starts = np.arange(0, n, B); counts = np.minimum(B, n - starts)
    alphas = [orc.adam_alpha(lr, i + 1) for i in range(len(starts))]
    arena = int(rng.choice([16, 32, 64]))
    use_graph = bool(rng.random() < 0.6)
    chunks, left = [], len(starts)
    while left > 0:
        k = int(min(left, rng.choice([1, 2, 3, 5, 8, 9, 16, 21, 33, 40])))
        chunks.append(k); left -= k
    out = {}
    for lazy in (False, True):
        eng = TrainEngine(n_u, n_a, max_batch=B, arena_steps=arena, lazy=lazy)
        assert eng.lazy == lazy
        eng.set_head(w=1.2); eng.set_weights(U, A)
        eng.set_epoch(ui, ai, t, starts, counts, alphas)
        done = 0
        for k in chunks:
            done += eng.run(k, use_graph=use_graph, first_step=done)
        eng.synchronize()
        out[lazy] = (eng.W.clone(), eng.M.clone(), eng.V.clone(), eng.read_state())
        eng.close()
    (W0, M0, V0, r0), (W1, M1, V1, r1) = out[False], out[True]
    ok = torch.equal(W0, W1) and torch.equal(M0, M1) and torch.equal(V0, V1)
    for key in ("w", "b", "gamma", "beta", "adam_m", "adam_v", "mov_mean", "mov_var", "se_sum", "n_seen", "step_fwd"):
        ok = ok and np.array_equal(r0[key], r1[key])
    ok = ok and abs(float(r0["last_loss"]) - float(r1["last_loss"])) <= 3e-6 * abs(float(r0["last_loss"])) + 1e-7
    if not ok:
        print("MISMATCH config", c, dict(n_u=n_u, n_a=n_a, B=B, steps=len(starts), lr=lr, arena=arena, graph=use_graph, chunks=chunks))
        sys.exit(1)
    if c % 5 == 4:
        print("config %d ok  %.0f s" % (c, time.time() - t0), flush=True)
print("soak ok: %d configs" % n_cfg)
