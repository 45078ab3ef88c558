"""k_cand time alone (HIP events around its launches, one chain) for one library build:
   [ANIREC_LIB_PATH=...] [ANIREC_TOPK_DEBUG=1] python scripts/cand_time.py n nq k [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anime_recommendations_amd import ops
n, nq, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
g = torch.Generator(device="cuda"); g.manual_seed(7)
Wh = ops.rownorm(torch.randn(n, 128, generator=g, device="cuda") * 0.05)
q = torch.arange(nq, dtype=torch.int32, device="cuda")
out = []
for r in range(rounds + 1):
    acc = {}
    ops.cosine_topk_mfma(Wh, q, k, prior=None, fallback=False, cand_timing=acc, batch=nq, lanes=1)
    if r: out.append(acc["ms"])
fl = 2.0 * nq * n * 128
print("%s dbg=%s n=%d nq=%d k=%d: k_cand ms %s  -> %.0f TFLOP/s (%d launches)" % (os.path.basename(os.environ.get("ANIREC_LIB_PATH", "libanirec.so")), os.environ.get("ANIREC_TOPK_DEBUG", "0"), n, nq, k, " ".join("%.3f" % x for x in out), fl / min(out) / 1e9, acc["launches"]))
