import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anime_recommendations_amd import ops
n_u, n_a, nq = 350000, 18000, int(sys.argv[1]) if len(sys.argv) > 1 else 100000
g = torch.Generator(device="cuda"); g.manual_seed(7)
U = torch.randn(n_u, 128, generator=g, device="cuda") * 0.05
A = torch.randn(n_a, 128, generator=g, device="cuda") * 0.05
head = dict(w=1.3, b=0.1, gamma=0.9, beta=-0.2, mov_mean=0.05, mov_var=0.4)
users = torch.arange(nq, dtype=torch.int32, device="cuda")
out = torch.empty(nq, n_a, dtype=torch.float32, device="cuda")
for name, fn in (("mfma", lambda: ops.predict_grid_mfma(U, A, head, users, out=out)),):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s predict %d x %d: %.2f ms  %.2f G ratings/s  write %.2f TB/s  %.1f TFLOP/s (1 product)" % (
        name, nq, n_a, dt * 1e3, nq * n_a / dt / 1e9, nq * n_a * 4 / dt / 1e12, 2.0 * nq * n_a * 128 / dt / 1e12))
