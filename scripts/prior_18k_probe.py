"""How much would a threshold prior buy the 18 k x 18 k top-k job?  The job's own k-th best scores give the ideal
prior (a low quantile of them minus the fp16 error window); time the job with it, count the rows it refutes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anime_recommendations_amd import ops
n, K = 18000, int(sys.argv[1]) if len(sys.argv) > 1 else 100
g = torch.Generator(device="cuda"); g.manual_seed(7)
Wh = ops.rownorm(torch.randn(n, 128, generator=g, device="cuda") * 0.05)
q = torch.arange(n, dtype=torch.int32, device="cuda")
idx0, sim0, _ = ops.cosine_topk_mfma(Wh, q, K)
kth = sim0[:, K - 1].float()
print("k-th best: min %.4f  q0.1%% %.4f  q1%% %.4f  median %.4f  max %.4f" % tuple(float(x) for x in (
    kth.min(), torch.quantile(kth, 0.001), torch.quantile(kth, 0.01), kth.median(), kth.max())))
cfgs = [("no prior", None)] + [("prior q%g" % qq, float(torch.quantile(kth, qq)) - 0.0101) for qq in (0.0, 0.001, 0.01, 0.1, 0.5)]
res = {c[0]: [] for c in cfgs}
info = {}
for r in range(6):
    for name, pr in cfgs:
        st = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        idx, sim, nfb = ops.cosine_topk_mfma(Wh, q, K, prior=pr, stats=st)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if r == 0:
            info[name] = (bool(torch.equal(idx, idx0)), int(nfb), st.get("rerun_rows"), pr)
        else:
            res[name].append(dt)
for name, v in res.items():
    t = np.array(v)
    print("%-12s prior %s: median %.3f ms  min %.3f ms  identical %s fallback %s rerun %s" % (
        name, "%.4f" % info[name][3] if info[name][3] is not None else "-", np.median(t) * 1e3, t.min() * 1e3,
        info[name][0], info[name][1], info[name][2]))
