"""Wall time of the all-pairs cosine top-k job with the library defaults: python scripts/time_topk_allpairs.py n k label"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anime_recommendations_amd import ops
n, k = int(sys.argv[1]), int(sys.argv[2])
g = torch.Generator(device="cuda"); g.manual_seed(7)
Wh = ops.rownorm(torch.randn(n, 128, generator=g, device="cuda") * 0.05)
q = torch.arange(n, dtype=torch.int32, device="cuda")
ts = []
for r in range(5):
    st = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = ops.cosine_topk_mfma(Wh, q, k, stats=st)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("%s: ms %s batches %d allpairs %s rerun %d" % (sys.argv[3], " ".join("%.2f" % t for t in ts[1:]), st["batches"], st["allpairs"], st["rerun_rows"]), flush=True)
