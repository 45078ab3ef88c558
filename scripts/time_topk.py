import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anime_recommendations_amd import ops
cfgs = ((18000, 18000), (350000, 65536)) if len(sys.argv) < 2 else ((int(sys.argv[1]), int(sys.argv[2])),)
K = int(sys.argv[3]) if len(sys.argv) > 3 else 100
for n, nq in cfgs:
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    W = torch.randn(n, 128, generator=g, device="cuda") * 0.05
    Wh = ops.rownorm(W)
    q = torch.arange(nq, dtype=torch.int32, device="cuda")
    fn = lambda: ops.cosine_topk_mfma(Wh, q, K, fallback=not os.environ.get('ANIREC_TOPK_DEBUG'))
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    fl = 2.0 * nq * n * 128
    print("mfma k=%d n=%d nq=%d: %.2f ms  %.1f TFLOP/s  %.0f queries/s  fallback=%d" % (K, n, nq, dt * 1e3, fl / dt / 1e12, nq / dt, out[2]))
    # the same measurement bench.py's roofline uses (HIP events around the k_cand launches, on their stream), in
    # THIS process: under `rocprofv3 --kernel-trace --stats` it can be held against the trace of the same call
    ops.topk_mfma_timing(True); fn(); ms, nl = ops.topk_mfma_timing(False)
    print("k_cand by HIP events, third call: %.3f ms over %d launches = %.1f TFLOP/s" % (ms, nl, fl / ms / 1e9))
