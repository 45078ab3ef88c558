import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from anime_recommendations_amd.engine import TrainEngine
from anime_recommendations_amd import schedule
dev = torch.device("cuda:0")
for nseg in (1, 2, 4, 8):
    n_users, n_anime, B = 350000 // nseg, 18000, 10000
    cap = min(16384, int(B + 6 * np.sqrt(B) + 16)) if nseg > 1 else B
    ui, ai, t = bench.synth_ratings(n_users, n_anime, 4 * B, dev)
    U, A = bench.init_tables(n_users, n_anime, dev)
    eng = TrainEngine(n_users, n_anime, max_batch=cap, arena_steps=8, n_seg=nseg, my_seg=0, anime_dense=nseg > 1)
    eng.set_head(w=1.2); eng.set_weights(U, A)
    eng.set_epoch(ui, ai, t, np.arange(4) * B, np.full(4, B), schedule.adam_alphas(1e-5, 1, 4))
    eng.prep(0, 4); eng.fwd(); eng.synchronize()
    pf = eng.packet_floats
    for s in range(1, nseg):
        eng.packets[s * pf:(s + 1) * pf] = eng.packets[:pf]
    torch.cuda.synchronize()
    evs = []
    for name in ("head", "bwd", "adam_users" if nseg > 1 else "adam", "adam_anime_finish" if nseg > 1 else None):
        if name is None: continue
        ts = []
        for rep in range(20):
            if name in ("adam", "adam_anime_finish"):
                pass
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(eng.stream); getattr(eng, name)(); e1.record(eng.stream); eng.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
            if name in ("adam", "adam_anime_finish"):
                # keep the cursor inside the schedule
                rec = eng.read_state(); rec["step_fwd"] = 0; eng.write_state(rec)
        evs.append((name, np.median(ts)))
    print("n_seg", nseg, "rows", n_users + n_anime, " ".join("%s=%.1fus" % e for e in evs))
    eng.close()
