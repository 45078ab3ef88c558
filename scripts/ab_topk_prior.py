"""All-pairs (or a slice) cosine top-k with and without the threshold prior (ANIREC_TOPK_PRIOR), interleaved rounds in
one process; checks that the lists are identical.  usage: ab_topk_prior.py [n] [nq] [k] [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anime_recommendations_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 350_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else n
k = int(sys.argv[3]) if len(sys.argv) > 3 else 100
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 2
g = torch.Generator(device="cuda"); g.manual_seed(7)
Wh = ops.rownorm(torch.randn(n, 128, generator=g, device="cuda") * 0.05)
q = torch.arange(nq, dtype=torch.int32, device="cuda")
ref = None
for r in range(rounds + 1):
    for pr in ("0", "1"):
        os.environ["ANIREC_TOPK_PRIOR"] = pr
        torch.cuda.synchronize(); t0 = time.perf_counter()
        idx, sim, nfb = ops.cosine_topk_mfma(Wh, q, k)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if ref is None:
            ref = (idx.clone(), sim.clone())
        same = bool(torch.equal(idx, ref[0]) and torch.equal(sim, ref[1]))
        if r:
            print("prior=%s: %.2f ms  %.2f M queries/s  exact-path rows=%d  identical=%s" % (pr, dt * 1e3, nq / dt / 1e6, nfb, same))
