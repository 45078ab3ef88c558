import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from anime_recommendations_amd.engine import TrainEngine
from anime_recommendations_amd import schedule
N_USERS, N_ANIME, B = 350_000, 18_000, 10_000
dev = torch.device("cuda:0")
def run(steps, use_graph=False):
    ui, ai, t = bench.synth_ratings(N_USERS, N_ANIME, steps * B, dev, seed=1)
    U, A = bench.init_tables(N_USERS, N_ANIME, dev)
    eng = TrainEngine(N_USERS, N_ANIME, max_batch=B, arena_steps=8)
    eng.set_head(w=1.2); eng.set_weights(U, A)
    eng.set_epoch(ui, ai, t, np.arange(steps) * B, np.full(steps, B), schedule.adam_alphas(1e-5, 1, steps))
    eng.run(steps, use_graph=use_graph); eng.synchronize()
    out = dict(ui=ui.cpu().numpy(), ai=ai.cpu().numpy(), t=t.cpu().numpy(), U0=U.cpu().numpy(), W=eng.W.cpu().numpy(),
               M=eng.M.cpu().numpy(), V=eng.V.cpu().numpy(), st=eng.read_state())
    eng.close(); return out
for steps in (1, 3):
    a, b = run(steps), run(steps)
    for k in ("ui", "ai", "t", "U0", "W", "M", "V"):
        d = a[k] != b[k]
        print(steps, k, "mismatch", int(d.sum()), "rows", np.unique(np.nonzero(d)[0])[:8] if d.ndim > 1 and d.any() else "")
    print(steps, "loss", a["st"]["last_loss"], b["st"]["last_loss"], a["st"]["bn_mu"], b["st"]["bn_mu"])
