"""Latency of the reference's literal call: ONE query row against all keys (similar_users.py:293-296)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anime_recommendations_amd import ops
for n in (18_000, 350_000):
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    W = torch.randn(n, 128, generator=g, device="cuda") * 0.05
    Wh = ops.rownorm(W)
    ws = None
    for nq in (1, 16, 256):
        q = torch.arange(nq, dtype=torch.int32, device="cuda") * 3
        for _ in range(3): ops.cosine_topk(Wh, q, 10)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 20
        for _ in range(reps): ops.cosine_topk(Wh, q, 10)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        print("exact cosine_topk n=%d nq=%d: %.1f us  (%.2f TB/s of key bytes)" % (n, nq, dt * 1e6, n * 512 / dt / 1e12))
