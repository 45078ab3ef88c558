"""Randomised equality soak of the all-pairs cosine job against the plain job (several sizes, k, seeds, data shapes):
   python scripts/soak_topk_allpairs.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anime_recommendations_amd import ops
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(2026)
bad = 0
for rnd in range(rounds):
    for n in (131072, 150001, 222222, 350000, 401408):
        for k in (8, 33, 100, 127):
            seed = int(rng.integers(1, 1 << 30))
            g = torch.Generator(device="cuda"); g.manual_seed(seed)
            shape = ("iso", "sub64", "clu300")[int(rng.integers(0, 3))]
            W = torch.randn(n, 128, generator=g, device="cuda")
            if shape == "sub64":
                W[:, 64:] *= 0.05
            elif shape == "clu300":
                c = torch.randn(n // 300, 128, generator=g, device="cuda")
                W = c[torch.randperm(n, generator=g, device="cuda") % (n // 300)] + 0.35 * W
            Wh = ops.rownorm(W)
            q = torch.arange(n, dtype=torch.int32, device="cuda")
            forced = bool(rng.integers(0, 2))
            st1, st0 = {}, {}
            t0 = time.perf_counter()
            i1, s1, f1 = ops.cosine_topk_mfma(Wh, q, k, stats=st1, allpairs=(True if forced else "auto"))
            torch.cuda.synchronize(); t1 = time.perf_counter()
            i0, s0, f0 = ops.cosine_topk_mfma(Wh, q, k, stats=st0, allpairs=False)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            ok = torch.equal(i0, i1) and torch.equal(s0, s1)
            bad += not ok
            print("n=%d k=%d %s seed=%d forced=%d: allpairs=%s %.1f ms (rerun %d fb %d) plain %.1f ms (rerun %d fb %d)  %s" % (
                n, k, shape, seed, forced, st1["allpairs"], (t1 - t0) * 1e3, st1["rerun_rows"], st1["fallback_rows"],
                (t2 - t1) * 1e3, st0["rerun_rows"], st0["fallback_rows"], "EQUAL" if ok else "DIFFERENT"), flush=True)
            del W, Wh, i0, i1, s0, s1
print("different:", bad)
sys.exit(1 if bad else 0)
