"""A/B of the train-step launch structures on one box, interleaved rounds in one process:
graph_serial (captured graph, one branch), graph_overlap (two graph branches), eager_overlap (two streams, no graph).
usage: time_train_modes.py [s109m|s7m] [steps] [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from anime_recommendations_amd.engine import TrainEngine

wl = sys.argv[1] if len(sys.argv) > 1 else "s109m"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 192
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
modes = sys.argv[4:] or ["graph_serial", "graph_overlap", "eager_overlap"]
n_users, n_anime = bench.WORKLOADS[wl]
B = 10_000
dev = torch.device("cuda:0")
total = steps * (rounds + 1)
ui, ai, t = bench.synth_ratings(n_users, n_anime, total * B, dev)
U, A = bench.init_tables(n_users, n_anime, dev)
engs = {}
for m in modes:
    os.environ["ANIREC_TRAIN_MODE"] = m
    e = TrainEngine(n_users, n_anime, max_batch=B, arena_steps=64)
    e.set_head(w=1.2)
    e.set_weights(U, A)
    e.set_epoch(ui, ai, t, np.arange(total) * B, np.full(total, B), bench.alphas_for(total))
    e.run(steps)            # warm-up: captures the graph in this mode
    e.synchronize()
    engs[m] = e
res = {m: [] for m in modes}
for r in range(rounds):
    for m in modes:
        os.environ["ANIREC_TRAIN_MODE"] = m
        e = engs[m]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.run(steps)
        e.synchronize()
        res[m].append((time.perf_counter() - t0) / steps * 1e6)
for m in modes:
    v = np.array(res[m])
    print("%-14s %s: median %.1f us/step  min %.1f" % (m, wl, np.median(v), v.min()))
ws = [engs[m].W.clone() for m in modes]
print("bitwise equal tables across modes:", all(torch.equal(ws[0], w) for w in ws[1:]))
