"""End-to-end time of trainer.fit (the model.fit replacement) on a synthetic table: where does an epoch go?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from anime_recommendations_amd import data, trainer

wl = sys.argv[1] if len(sys.argv) > 1 else "s7m"
n_users, n_anime = bench.WORKLOADS[wl]
n = int(sys.argv[2]) if len(sys.argv) > 2 else (7_000_000 if wl == "s7m" else 30_000_000)
ui, ai, t = bench.synth_ratings(n_users, n_anime, n, torch.device("cuda"))
table = data.RatingTable(ui.cpu().numpy().astype(np.int64), ai.cpu().numpy().astype(np.int64),
                         t.cpu().numpy().astype(np.float64), np.arange(n_users), np.arange(n_anime))
del ui, ai, t
cfg = trainer.FitConfig(epochs=int(sys.argv[3]) if len(sys.argv) > 3 else 4, batch_size=10_000, test_size=10_000, verbose=1)
marks = []
def log(msg):
    marks.append(time.perf_counter()); print(msg)
t0 = time.perf_counter()
res = trainer.fit(table, cfg, log=log)
t1 = time.perf_counter()
steps = (n - 10_000 + 9_999) // 10_000
print("fit: %d ratings, %d steps/epoch: total %.2f s; first epoch mark %.2f s, later epochs %s s; pure GPU step time would be %.3f s/epoch" % (
    n, steps, t1 - t0, marks[0] - t0, ["%.3f" % (b - a) for a, b in zip(marks, marks[1:])], steps * (0.0455 if wl == "s7m" else 0.216) * 1e-3))
