"""model.predict grid at 100 k x 18 k: the dword-store kernel (ANIREC_PREDICT_KERNEL=1) against the row-quad /
pipelined-epilogue kernel, interleaved rounds in ONE process (cdna_hip_programming.md §5.4 rule 24)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anime_recommendations_amd import ops
n_u, n_a, nq = 350000, 18000, int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
g = torch.Generator(device="cuda"); g.manual_seed(7)
U = torch.randn(n_u, 128, generator=g, device="cuda") * 0.05
A = torch.randn(n_a, 128, generator=g, device="cuda") * 0.05
head = dict(w=1.3, b=0.1, gamma=0.9, beta=-0.2, mov_mean=0.05, mov_var=0.4)
users = torch.arange(nq, dtype=torch.int32, device="cuda")
out = torch.empty(nq, n_a, dtype=torch.float32, device="cuda")
variants = (("dword stores (v1)", "1"), ("row-quad nt stores + pipelined epilogue (v2)", "0")) if rounds > 1 else (("v2", "0"),)
res = {k: [] for k, _ in variants}
outs = {}
for r in range(rounds + 1):
    for name, env in variants:
        os.environ["ANIREC_PREDICT_KERNEL"] = env
        torch.cuda.synchronize()
        t0 = time.perf_counter(); ops.predict_grid_mfma(U, A, head, users, out=out); torch.cuda.synchronize()
        if r:
            res[name].append(time.perf_counter() - t0)
        elif nq <= 20000:
            outs[name] = out.clone()
for name, v in res.items():
    v = np.array(v)
    print("%-46s %d x %d: median %.3f ms  min %.3f ms  -> %.2f TB/s written (median)" % (
        name, nq, n_a, np.median(v) * 1e3, v.min() * 1e3, nq * n_a * 4 / np.median(v) / 1e12))
if len(outs) == 2:
    a, b = list(outs.values())
    print("max |v1 - v2| = %.3g" % float((a - b).abs().max()))
