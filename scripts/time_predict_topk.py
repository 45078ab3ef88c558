"""Time model_recs' batched top-k (predict + unwatched mask + top-k) at C5: 100 k users x 18 k anime."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anime_recommendations_amd import ops

n_u, n_a, nq, k = 350_000, 18_000, int(sys.argv[1]) if len(sys.argv) > 1 else 100_000, 10
g = torch.Generator(device="cuda"); g.manual_seed(7)
U = torch.randn(n_u, 128, generator=g, device="cuda") * 0.05
A = torch.randn(n_a, 128, generator=g, device="cuda") * 0.05
head = dict(w=1.3, b=0.1, gamma=0.9, beta=-0.2, mov_mean=0.05, mov_var=0.4)
users = torch.arange(nq, dtype=torch.int32, device="cuda")
watched = torch.randint(-2**31, 2**31 - 1, (nq, (n_a + 31) // 32), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
watched &= torch.randint(-2**31, 2**31 - 1, watched.shape, generator=g, device="cuda", dtype=torch.int64).to(torch.int32)  # ~25 % watched
for name, fn in (("predict_topk (exact fp32 path)", lambda: ops.predict_topk(U, A, head, users, k, watched)),
                 ("predict_topk_mfma", lambda: ops.predict_topk_mfma(U, A, head, users, k, watched))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s nq=%d: %.2f ms  %.2f M users/s  %.1f G ratings/s%s" % (name, nq, dt * 1e3, nq / dt / 1e6, nq * n_a / dt / 1e9, ("  fallback=%d" % out[2]) if len(out) > 2 else ""))
