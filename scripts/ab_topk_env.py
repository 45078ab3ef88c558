"""A/B of an environment knob of the top-k job in interleaved rounds of one process:
   python scripts/ab_topk_env.py n nq k ENVNAME v1,v2,... [rounds]   (value '-' = unset)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anime_recommendations_amd import ops
n, nq, k, name = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
vals = sys.argv[5].split(",")
rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 3
g = torch.Generator(device="cuda"); g.manual_seed(7)
Wh = ops.rownorm(torch.randn(n, 128, generator=g, device="cuda") * 0.05)
q = torch.arange(nq, dtype=torch.int32, device="cuda")
res, ref, sts = {}, None, {}
for rnd in range(rounds + 1):
    for v in vals:
        if v == "-": os.environ.pop(name, None)
        else: os.environ[name] = v
        reps = 2 if nq > 100000 else 8
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            st = {}
            out = ops.cosine_topk_mfma(Wh, q, k, stats=st)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        if ref is None: ref = (out[0].clone(), out[1].clone())
        assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
        if rnd: res.setdefault(v, []).append(dt * 1e3)
        sts[v] = (st["rerun_rows"], st["fallback_rows"])
for v, t in res.items():
    print("n=%d nq=%d k=%d %s=%s: ms %s  rerun/fallback %s" % (n, nq, k, name, v, " ".join("%.3f" % x for x in t), sts[v]), flush=True)
