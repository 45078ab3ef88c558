"""model.predict grid at 100 k x 18 k: variants of the kernel timed in interleaved rounds in ONE process
(cdna_hip_programming.md §5.4 rule 24).  usage: time_predict.py [n_users] [rounds] [variant ...]
variants: v1 (dword stores), v2 (row-quad nt stores + pipelined epilogue), v2p1 (v2 without the anime-part grid),
v2dbg1 (stores without MFMAs), v2dbg2 (MFMAs without stores)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anime_recommendations_amd import ops
n_u, n_a, nq = 350000, 18000, int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ENVS = {"v1": dict(ANIREC_PREDICT_KERNEL="1"), "v2": {}, "v2p1": dict(ANIREC_PREDICT_PARTS="1"),
        "v2p3": dict(ANIREC_PREDICT_PARTS="3"), "v2p12": dict(ANIREC_PREDICT_PARTS="12"),
        "v2dbg1": dict(ANIREC_PREDICT_DEBUG="1"), "v2dbg2": dict(ANIREC_PREDICT_DEBUG="2")}
variants = sys.argv[3:] or ["v1", "v2"]
g = torch.Generator(device="cuda"); g.manual_seed(7)
U = torch.randn(n_u, 128, generator=g, device="cuda") * 0.05
A = torch.randn(n_a, 128, generator=g, device="cuda") * 0.05
head = dict(w=1.3, b=0.1, gamma=0.9, beta=-0.2, mov_mean=0.05, mov_var=0.4)
users = torch.arange(nq, dtype=torch.int32, device="cuda")
out = torch.empty(nq, n_a, dtype=torch.float32, device="cuda")
res = {k: [] for k in variants}
outs = {}
for r in range(rounds + 1):
    for name in variants:
        for k in ("ANIREC_PREDICT_KERNEL", "ANIREC_PREDICT_PARTS", "ANIREC_PREDICT_DEBUG"):
            os.environ.pop(k, None)
        os.environ.update(ENVS[name])
        torch.cuda.synchronize()
        t0 = time.perf_counter(); ops.predict_grid_mfma(U, A, head, users, out=out); torch.cuda.synchronize()
        if r:
            res[name].append(time.perf_counter() - t0)
        elif nq <= 20000 and "dbg" not in name:
            outs[name] = out.clone()
for name, v in res.items():
    v = np.array(v)
    print("%-8s %d x %d: median %.3f ms  min %.3f ms  -> %.2f TB/s written (median)" % (
        name, nq, n_a, np.median(v) * 1e3, v.min() * 1e3, nq * n_a * 4 / np.median(v) / 1e12))
ks = list(outs)
for k in ks[1:]:
    print("max |%s - %s| = %.3g" % (ks[0], k, float((outs[ks[0]] - outs[k]).abs().max())))
