"""18 k x 18 k top-100 job (BASELINE configs[3], anime leg) under schedule knobs, interleaved rounds in one process:
whole-job time (rownorm excluded), fallback rows, lists compared with the default schedule's.
usage: sweep_topk_18k.py [k]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anime_recommendations_amd import ops
n, K = 18000, int(sys.argv[1]) if len(sys.argv) > 1 else 100
g = torch.Generator(device="cuda"); g.manual_seed(7)
Wh = ops.rownorm(torch.randn(n, 128, generator=g, device="cuda") * 0.05)
q = torch.arange(n, dtype=torch.int32, device="cuda")
CFG = {"default": {}}
for gr in (150, 200, 300, 400):
    CFG["growth%d" % gr] = dict(ANIREC_TOPK_GROWTH=str(gr))
for sp in (1, 2, 3, 4):
    CFG["splits%d" % sp] = dict(ANIREC_TOPK_SPLITS=str(sp))
CFG["growth200splits4"] = dict(ANIREC_TOPK_GROWTH="200", ANIREC_TOPK_SPLITS="4")
CFG["growth300splits4"] = dict(ANIREC_TOPK_GROWTH="300", ANIREC_TOPK_SPLITS="4")
KEYS = ("ANIREC_TOPK_GROWTH", "ANIREC_TOPK_SPLITS")
res = {k: [] for k in CFG}
ref = None
for r in range(6):
    for name, env in CFG.items():
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        idx, sim, nfb = ops.cosine_topk_mfma(Wh, q, K)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if r == 0:
            if ref is None:
                ref = (idx.clone(), sim.clone())
            same = bool(torch.equal(idx, ref[0]) and torch.equal(sim, ref[1]))
            res[name].append((same, int(nfb)))
        else:
            res[name].append(dt)
for name, v in res.items():
    t = np.array(v[1:])
    print("%-20s median %.3f ms  min %.3f ms  lists identical %s  fallback rows %d" % (name, np.median(t) * 1e3, t.min() * 1e3, v[0][0], v[0][1]))
