"""A/B of the job's chain count (ops.cosine_topk_mfma lanes=1..3) in interleaved rounds of ONE process:
   python scripts/ab_topk_lanes.py [n nq k]   (default: the 350 k x 350 k and the 18 k x 18 k top-100 jobs)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anime_recommendations_amd import ops
cfgs = ((350000, 350000, 100), (18000, 18000, 100), (350000, 350000, 10)) if len(sys.argv) < 4 else ((int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])),)
for n, nq, k in cfgs:
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    Wh = ops.rownorm(torch.randn(n, 128, generator=g, device="cuda") * 0.05)
    q = torch.arange(nq, dtype=torch.int32, device="cuda")
    ref = None
    res = {}
    for rnd in range(4):
        for lanes in (1, 2, 3):
            ops.cosine_topk_mfma(Wh, q, k, lanes=lanes) if rnd == 0 else None
            torch.cuda.synchronize(); t0 = time.perf_counter()
            reps = 2 if nq > 100000 else 10
            for _ in range(reps):
                st = {}
                out = ops.cosine_topk_mfma(Wh, q, k, lanes=lanes, stats=st)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
            if ref is None: ref = (out[0].clone(), out[1].clone())
            assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
            res.setdefault(lanes, []).append(dt * 1e3)
    for lanes, v in res.items():
        print("n=%d nq=%d k=%d lanes=%d: ms per call %s  (rerun %d fallback %d batches %d)" % (n, nq, k, lanes, " ".join("%.3f" % x for x in v), st["rerun_rows"], st["fallback_rows"], st["batches"]), flush=True)
