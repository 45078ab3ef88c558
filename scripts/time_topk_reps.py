"""Wall time of the cosine top-k job, several repetitions in one process (for A/B of library builds through
ANIREC_LIB_PATH):  python scripts/time_topk_reps.py n nq k [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anime_recommendations_amd import ops
n, nq, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
g = torch.Generator(device="cuda"); g.manual_seed(7)
Wh = ops.rownorm(torch.randn(n, 128, generator=g, device="cuda") * 0.05)
q = torch.arange(nq, dtype=torch.int32, device="cuda")
ops.cosine_topk_mfma(Wh, q, k); torch.cuda.synchronize()
ts = []
for _ in range(reps):
    inner = 1 if nq > 100000 else 20
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(inner):
        out = ops.cosine_topk_mfma(Wh, q, k)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / inner * 1e3)
print("lib=%s n=%d nq=%d k=%d ms: %s  checksum %d" % (os.path.basename(os.environ.get("ANIREC_LIB_PATH", "tree")), n, nq, k,
      " ".join("%.3f" % t for t in ts), int(out[0].to(torch.int64).sum())), flush=True)
