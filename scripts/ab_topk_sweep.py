"""Sweep of the job's chain count / stagger on one box, interleaved rounds in one process per library build:
   python scripts/ab_topk_sweep.py n nq k "lanes:stagger,lanes:stagger,..." [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anime_recommendations_amd import ops
n, nq, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfgs = [tuple(int(x) for x in c.split(":")) for c in sys.argv[4].split(",")]
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 3
g = torch.Generator(device="cuda"); g.manual_seed(7)
Wh = ops.rownorm(torch.randn(n, 128, generator=g, device="cuda") * 0.05)
q = torch.arange(nq, dtype=torch.int32, device="cuda")
dbg = bool(os.environ.get("ANIREC_TOPK_DEBUG"))
res = {}
ref = None
for rnd in range(rounds + 1):
    for lanes, stg in cfgs:
        os.environ["ANIREC_TOPK_STAGGER_PCT"] = str(stg % 1000); os.environ["ANIREC_TOPK_SIDE_WAVES"] = str(stg // 1000)
        reps = 2 if nq > 100000 else 8
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            st = {}
            out = ops.cosine_topk_mfma(Wh, q, k, lanes=lanes, stats=st, fallback=not dbg)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        if not dbg:
            if ref is None: ref = (out[0].clone(), out[1].clone())
            assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
        if rnd: res.setdefault((lanes, stg), []).append(dt * 1e3)
for (lanes, stg), v in res.items():
    print("n=%d nq=%d k=%d lanes=%d stagger=%d%%: ms %s" % (n, nq, k, lanes, stg, " ".join("%.3f" % x for x in v)), flush=True)
