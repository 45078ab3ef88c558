"""All-pairs cosine top-k: the symmetric schedule (prior_mode 3) against the plain job, same process.
   python scripts/ab_topk_sym.py n k [reps] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anime_recommendations_amd import ops
n, k = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
batch = int(sys.argv[4]) if len(sys.argv) > 4 else None
g = torch.Generator(device="cuda"); g.manual_seed(7)
Wh = ops.rownorm(torch.randn(n, 128, generator=g, device="cuda") * 0.05)
q = torch.arange(n, dtype=torch.int32, device="cuda")
res = {}
for name, ap in (("plain", False), ("allpairs", "auto")):
    ts = []
    for r in range(reps + 1):
        st = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = ops.cosine_topk_mfma(Wh, q, k, stats=st, allpairs=ap, batch=batch)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    res[name] = out
    print("%s n=%d k=%d: ms %s  batches %d lanes %d allpairs %s rerun %d fallback %d" % (
        name, n, k, " ".join("%.3f" % t for t in ts[1:]), st["batches"], st["lanes"], st["allpairs"], st["rerun_rows"],
        st["fallback_rows"]), flush=True)
a, b = res["plain"], res["allpairs"]
print("idx equal:", bool(torch.equal(a[0], b[0])), " score equal:", bool(torch.equal(a[1], b[1])))
if not torch.equal(a[0], b[0]):
    bad = torch.nonzero((a[0] != b[0]).any(1)).flatten()
    print("rows differing:", bad.numel(), bad[:10].tolist())
    r = int(bad[0]); print(a[0][r][:12].tolist(), b[0][r][:12].tolist()); print(a[1][r][:6].tolist(), b[1][r][:6].tolist())
