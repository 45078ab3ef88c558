#!/usr/bin/env python
"""bench.py — training ratings/sec of the MI355X hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" is one optimiser step of the reference's model.fit (neural_network.py:210-217) over
one batch of B = 10 000 ratings PER GPU (weak scaling: config.yaml:59 batch_size, scaled by
the replica count as the reference's TPU branch intends, neural_network.py:176-177), on the
synthetic S109M table shape (350 000 users x 18 000 anime, D = 128) — BASELINE.json configs[2],
the configuration the 1/2/4/8-GPU metric is quoted on; the 7M-rating shape of configs[1]
(15 000 x 17 560) is measured in the same run and reported under "also".

Inputs (tables, Adam moments, the synthetic ratings, the schedule) are resident in HBM when
the timed region starts; the timed region contains everything a step needs: batch sort/chunk
prep, fwd, head, bwd, dense Adam — K steps, barrier + synchronize on both sides, max over ranks.

Prints ONE JSON line (rank 0) with `roofline` (the dominant kernel of the timed path on the bytes it moves: the lazy
flush at the S109M shape — VALU-bound, counters under profiles/ — or the dense fused Adam) and `cpu_baseline` (the
plain-C oracle port of the reference's CPU TensorFlow step, timed on this box's host cores; N=1 only).
"""
import argparse
import json
import os
import sys
import time

# the CPU baselines run OpenMP on every host core; idle workers must sleep, not spin, or they starve the
# Python thread that launches the next GPU leg (measured: a 5.6 ms leg read 15 ms right after a baseline)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("GOMP_SPINCOUNT", "0")
# HIP multiplexes a process's streams onto 4 hardware queues by default; this process creates more than a dozen
# (one per training engine, the cosine job's side streams), and when the job's two chains land on the SAME queue
# they serialise: 28.0 ms instead of 24.9 for the 350 k x 350 k top-100 job (measured in this file, both ways).
# Read by the runtime when it initialises, i.e. before the first torch.cuda call below.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (n_users, n_anime)  — SURVEY.md §8(d)
    "s109m": (350_000, 18_000),
    "s7m": (15_000, 17_560),
}
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ADAM_BYTES_PER_ELEM = 28       # read W,m,v,g + write W,m,v
FWD_BYTES_PER_RATING = 1048
BWD_BYTES_PER_RATING = 2060
RATING_PMF = [0.18, 0.005, 0.005, 0.01, 0.02, 0.05, 0.10, 0.20, 0.22, 0.13, 0.08]


def synth_ratings(n_users, n_anime, n, device, seed=20260101):
    """i.i.d. draws from the S7M/S109M marginal laws of SURVEY.md §8(d): users weighted by a
    lognormal(ln 311, 0.35) activity, anime Zipf(s=1) over a fixed permutation, MAL-like ratings.
    Inverse-CDF sampling (host fp64 CDFs + seeded device uniforms): bit-reproducible across runs and
    ranks, unlike torch.multinomial on the GPU."""
    import torch
    rng = np.random.Generator(np.random.PCG64(seed))
    act = np.exp(rng.standard_normal(n_users) * 0.35 + np.log(311.0))
    perm = torch.from_numpy(rng.permutation(n_anime)).to(device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)

    def draw(weights):
        cdf = np.cumsum(np.asarray(weights, np.float64))
        cdf = torch.from_numpy(cdf / cdf[-1]).to(device)
        u = torch.rand(n, generator=g, device=device, dtype=torch.float64)
        return torch.searchsorted(cdf, u).clamp_(max=len(weights) - 1)

    ui = draw(act)
    ai = perm[draw(1.0 / np.arange(1, n_anime + 1))]
    t = draw(RATING_PMF).to(torch.float32) / 10.0
    return ui.to(torch.int32), ai.to(torch.int32), t


def init_tables(n_users, n_anime, device, seed=7):
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    U = (torch.rand(n_users, 128, generator=g, device=device) - 0.5) * 0.1   # Keras 'uniform' init
    A = (torch.rand(n_anime, 128, generator=g, device=device) - 0.5) * 0.1
    return U, A


def alphas_for(n_steps, lr=1e-5, t0=0):
    from anime_recommendations_amd.schedule import adam_alpha
    return [adam_alpha(lr, t0 + i + 1) for i in range(n_steps)]


def run_single(workload, steps, warmup, batch, use_graph=True, cpu_baseline=True, quiet=False):
    import torch
    from anime_recommendations_amd.engine import TrainEngine
    n_users, n_anime = WORKLOADS[workload]
    dev = torch.device("cuda:0")
    total_steps = warmup + steps + 64         # timed region + instrumented per-kernel passes
    ui, ai, t = synth_ratings(n_users, n_anime, total_steps * batch, dev)
    U, A = init_tables(n_users, n_anime, dev)
    eng = TrainEngine(n_users, n_anime, max_batch=batch, arena_steps=64)
    eng.set_head(w=1.2)
    eng.set_weights(U, A)
    starts = np.arange(total_steps) * batch
    eng.set_epoch(ui, ai, t, starts, np.full(total_steps, batch), alphas_for(total_steps))
    if warmup:
        eng.run(warmup, use_graph=use_graph, first_step=0)
    eng.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run(steps, use_graph=use_graph, first_step=warmup)
    eng.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rec = eng.read_state()
    assert int(rec["step_fwd"]) == warmup + steps and np.isfinite(rec["last_loss"])

    # instrumented pass: the SAME step sequence continued eagerly, every kernel stamping the constant-clock time of each
    # workgroup's first and last instruction (anirec_train_stage_ticks): a launch's duration is max(end) - min(start),
    # taken in the step, on the batch the step really reads, behind whatever the previous kernel left in the caches
    # (round 2 replayed one batch 50 x: its gathered rows were cache-resident from the second repetition on).
    first = warmup + steps
    lazy_ms = None
    n_lazy = 0
    eng.stage_ticks(True, read=False)
    if eng.lazy:          # the path the timed region ran: 32 steps = four 8-step windows
        n_lazy = 32
        eng.run(n_lazy, use_graph=False, first_step=first)
        tk = eng.stage_ticks(True)
        lazy_ms = {k: tk[k] * 1e-3 for k in ("lazy_catchup", "fwd", "head", "bwd", "lazy_adam", "lazy_flush", "lazy_reduce")}
        lazy_ms["launches"] = {k: tk["launches"][k] for k in ("lazy_catchup", "lazy_adam", "lazy_flush", "lazy_reduce")}
    # the dense kernels, stage by stage, by the same stamps ...
    for k in range(8):
        eng.prep(first + n_lazy + k, 1)
        eng.fwd()
        eng.head()
        eng.bwd()
        eng.adam()
    tk = eng.stage_ticks(False)
    # ... and the dense Adam by HIP events on the engine's stream (the roofline contract), stamps OFF: the host is far
    # ahead of a 190 us kernel, so the event pair brackets the kernel and nothing else
    evs = []
    for k in range(16):
        eng.prep(first + n_lazy + 8 + k, 1)
        eng.fwd()
        eng.head()
        eng.bwd()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record(eng.stream)
        eng.adam()
        e1.record(eng.stream)
        evs.append((e0, e1))
    eng.synchronize()
    kern_ms = {k: tk[k] * 1e-3 for k in ("fwd", "head", "bwd")}
    kern_ms["adam_by_stamps"] = tk["adam"] * 1e-3
    kern_ms["adam"] = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    rows = n_users + n_anime
    adam_bytes = ADAM_BYTES_PER_ELEM * rows * 128
    adam_gbs = adam_bytes / (kern_ms["adam"] * 1e-3) / 1e9
    # HBM traffic from the PMC passes committed under profiles/ (PMC cannot be collected from inside this process);
    # quoted only for the workload it was measured on AND only while the kernel source is still the one that was
    # measured (git blob hash recorded next to the counters)
    tag = {"s109m": "train_s109m", "s7m": "train_s7m"}.get(workload) if batch == 10_000 else None
    dense_name = "k_adam<true>" if rows * 128 * 4 * 3 > (192 << 20) else "k_adam<false>"
    dense_roof = {"kernel": "k_adam (dense fused Adam, both tables: one step per launch)", "bound": "hbm",
                  "achieved": adam_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": adam_gbs / HBM_PEAK_GBS,
                  "traffic": pmc_traffic(tag, dense_name, source="anirec_train.hip") if tag else None,
                  "algorithmic_bytes_per_launch": adam_bytes, "avg_launch_ms": kern_ms["adam"]}
    step_bytes = (FWD_BYTES_PER_RATING + BWD_BYTES_PER_RATING) * batch + adam_bytes
    step_ms = dt / steps * 1e3
    out = {
        "value": steps * batch / dt,
        "ms_per_step": step_ms,
        "loss": float(rec["last_loss"]),
        "adam": "lazy" if eng.lazy else "dense",
        "kernels_ms": kern_ms,
        # whole-step roofline of SURVEY.md §8(d): (3.1 KB x B + 28 B x table elements) / step time
        "step_roofline": {"bound": "hbm", "algorithmic_bytes_per_step": step_bytes,
                          "achieved": step_bytes / (dt / steps) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": step_bytes / (dt / steps) / 1e9 / HBM_PEAK_GBS},
        "fwd_gbs": FWD_BYTES_PER_RATING * batch / (kern_ms["fwd"] * 1e-3) / 1e9,
        "bwd_gbs": BWD_BYTES_PER_RATING * batch / (kern_ms["bwd"] * 1e-3) / 1e9,
    }
    if eng.lazy:
        # The dominant kernel of the lazy path: ONE pass over W, M, V applies every row's pending steps of an 8-step
        # window.  `roofline` is on the bytes the pass MOVES (W, M, V read and written once: 24 B per element — the
        # kernel's own algorithmic bytes; PMC traffic agrees to 1 %), so frac <= 1 by construction.  It is not what
        # binds the kernel: the PMC passes (profiles/r04_pmc_valu_train_s109m.json) show its vector ALUs issuing 0.92
        # of the kernel's cycles (`valu`), i.e. it is bound by the IEEE sqrt + divide arithmetic the bit-exact replay
        # of Keras' Adam requires.  The ratio to SURVEY 8(d)'s dense-algorithm bytes (28 B x elements x steps covered)
        # is a separate key, `vs_dense_algorithm`: > 1 says how far under the dense update's traffic the window runs.
        win = lazy_ms["launches"]["lazy_adam"] / max(1, lazy_ms["launches"]["lazy_flush"])
        fl_ms = lazy_ms["lazy_flush"]
        alg = adam_bytes * win
        moved = 24 * rows * 128
        flush_name = "k_lazy_flush<true>" if dense_name == "k_adam<true>" else "k_lazy_flush<false>"
        out["lazy_kernels_ms"] = lazy_ms
        out["roofline"] = {"kernel": "k_lazy_flush (lazy dense Adam: every row's pending steps of a %d-step window in one "
                                     "pass over W, M, V; IEEE sqrt + divide per element-step)" % round(win),
                           "bound": "hbm", "achieved": moved / (fl_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": moved / (fl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "traffic": pmc_traffic(tag, flush_name, source="anirec_train.hip") if tag else None,
                           "algorithmic_bytes_per_launch": moved, "avg_launch_ms": fl_ms, "steps_per_launch": win,
                           "limiter": "valu", "valu": pmc_valu(tag, flush_name) if tag else None,
                           "vs_dense_algorithm": {"dense_bytes_for_the_steps_covered": alg,
                                                  "equivalent_gbs": alg / (fl_ms * 1e-3) / 1e9,
                                                  "ratio_to_hbm_peak": alg / (fl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                           "note": "achieved / frac: the 24 B per element the pass really moves over the kernel's mean "
                                   "duration (in-kernel stamps over the instrumented steps).  The kernel is VALU-bound "
                                   "(`valu`: counters of the committed PMC passes, null when the kernel source has "
                                   "changed since): the replay performs the dense kernel's fp32 operations — tables, Adam "
                                   "moments and scalar state are bit-identical to the dense path "
                                   "(dense_kernel_roofline) after every run() call, timed region included."}
        per_step = (lazy_ms["fwd"] + lazy_ms["head"] + lazy_ms["bwd"] + lazy_ms["lazy_adam"]
                    + (fl_ms + lazy_ms["lazy_reduce"]) / win)
        # what the lazy step really moves: the gathers, ~15 000 touched rows x (W, M, V) read + written twice (catch-up,
        # sparse step), and 1/8 of the flush's pass
        touched = 2 * 15_000 * 128 * 4 * 3 * 2
        moved_step = (FWD_BYTES_PER_RATING + BWD_BYTES_PER_RATING) * batch + touched + moved / win
        dense_alg = out["step_roofline"]
        out["step_roofline"] = {"bound": "hbm", "moved_bytes_per_step_estimate": moved_step,
                                "achieved": moved_step / (dt / steps) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": moved_step / (dt / steps) / 1e9 / HBM_PEAK_GBS,
                                "sum_of_kernels_ms": per_step,
                                "vs_dense_algorithm": {"algorithmic_bytes_per_step": dense_alg["algorithmic_bytes_per_step"],
                                                       "equivalent_gbs": dense_alg["achieved"],
                                                       "ratio_to_hbm_peak": dense_alg["frac"]},
                                "note": "frac is on the bytes the lazy step moves (gathers + touched rows twice + 1/8 of the "
                                        "flush pass); vs_dense_algorithm is the same step time on SURVEY 8(d)'s dense bytes"}
        out["dense_kernel_roofline"] = dense_roof
    else:
        out["roofline"] = dense_roof
        out["step_roofline"]["sum_of_kernels_ms"] = kern_ms["fwd"] + kern_ms["head"] + kern_ms["bwd"] + kern_ms["adam"]
    eng.close()
    del eng
    torch.cuda.empty_cache()
    if cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_port(n_users, n_anime, batch)
    return out


def cpu_baseline_port(n_users, n_anime, batch, budget_s=12.0):
    """The reference's CPU TensorFlow step restated in C + OpenMP (oracle/anirec_oracle.c),
    same table shape and batch size, timed on this box's host cores on a bounded sample."""
    from oracle import anirec_oracle as orc
    from oracle import c_oracle
    rng = np.random.default_rng(7)
    U = rng.uniform(-0.05, 0.05, (n_users, 128)).astype(np.float32)
    A = rng.uniform(-0.05, 0.05, (n_anime, 128)).astype(np.float32)
    st = orc.new_state(U, A, orc.new_head(w=1.2))
    max_steps = 400
    ui = rng.integers(0, n_users, max_steps * batch).astype(np.int32)
    ranks = 1.0 / np.arange(1, n_anime + 1)
    ai = rng.choice(n_anime, size=max_steps * batch, p=ranks / ranks.sum()).astype(np.int32)
    t = (rng.choice(11, size=max_steps * batch, p=np.array(RATING_PMF) / sum(RATING_PMF)) / 10).astype(np.float32)
    al = np.array([orc.adam_alpha(1e-5, i + 1) for i in range(max_steps)], np.float32)
    t0 = time.perf_counter()
    c_oracle.train_run(st, ui[:2 * batch], ai[:2 * batch], t[:2 * batch], batch, al[:2])
    per = (time.perf_counter() - t0) / 2
    n = int(max(3, min(max_steps - 2, budget_s / max(per, 1e-4))))
    t0 = time.perf_counter()
    c_oracle.train_run(st, ui[2 * batch:(2 + n) * batch], ai[2 * batch:(2 + n) * batch],
                       t[2 * batch:(2 + n) * batch], batch, al[2:2 + n])
    dt = time.perf_counter() - t0
    return {"value": n * batch / dt, "unit": "ratings/s", "cores": c_oracle.max_threads(), "kind": "port",
            "sample": "%d steps of %d ratings, %dx%d tables (plain-C/OpenMP restatement of the Keras "
                      "CPU step; TensorFlow 2.12 is not installable offline)" % (n, batch, n_users, n_anime)}


MFMA_F16_PEAK_TFLOPS = 2500.0   # dense f16 (= bf16) MFMA peak (MI355X_MICROARCH.md); the kernels use v_mfma_f32_16x16x32_f16


PMC_ROUND = "r04"


def git_blob_hash(path):
    """`git hash-object` of a file: sha1 over "blob <size>\\0" + content."""
    import hashlib
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def pmc_traffic(name, kernels=None, source=None, calls=None):
    """HBM bytes from the committed PMC passes (profiles/<round>_pmc_traffic_<name>.json): per-launch mean of one
    kernel, or the sum over the listed kernels of (mean bytes x launches per call).  The JSON records the git blob
    hash of every csrc/*.hip and of the host modules that shape the launches (ops.py ...) at collection time; if any
    file in `source` has changed since, the counters no longer describe what is being timed and None is returned."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "%s_pmc_traffic_%s.json" % (PMC_ROUND, name))))
        for src in ([source] if isinstance(source, str) else (source or [])):
            sub = "csrc" if src.endswith((".hip", ".hpp")) else ""
            cur = git_blob_hash(os.path.join(ROOT, "anime_recommendations_amd", sub, src))
            if rec.get("sources", {}).get(src) != cur:
                return None
        d = rec["kernels"]
        if isinstance(kernels, str):
            return d[kernels]["total_bytes_per_launch"]
        if calls:      # all launches of the listed kernels in one call of the op (the profiled script makes `calls` calls)
            return sum(d[k]["total_bytes_all_launches"] for k in kernels) / calls
        return sum(d[k]["total_bytes_per_launch"] * n for k, n in kernels.items())
    except (OSError, KeyError, ValueError, TypeError):
        return None


def pmc_valu(name, kernel):
    """VALU evidence of one kernel from the committed SQ counter passes (profiles/<round>_pmc_valu_<name>.json, written
    by scripts/pmc_valu_summary.py from three separate rocprofv3 --pmc passes): the share of the kernel's cycles its
    vector ALUs were issuing, the shares of a wave's resident time spent issuing VALU work / parked on memory / stalled
    at issue, the instruction count.  None when the kernel source has changed since the passes."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "%s_pmc_valu_%s.json" % (PMC_ROUND, name))))
        cur = git_blob_hash(os.path.join(ROOT, "anime_recommendations_amd", "csrc", "anirec_train.hip"))
        if rec.get("sources", {}).get("anirec_train.hip") != cur:
            return None
        k = rec["kernels"][kernel]
        return {x: k[x] for x in ("valu_pipe_busy", "valu_active_share", "wait_any_share", "wait_inst_share",
                                  "valu_insts_per_wave", "effective_clock_ghz", "avg_us_under_pmc") if x in k}
    except (OSError, KeyError, ValueError, TypeError):
        return None


def _cosine_leg(W, nq, k, cpu_baseline, reps=3, slice_note=None, traffic_name=None):
    """row-normalise + cosine top-k of `nq` query rows (rows 0..nq-1) against every row of W; queries/s of the whole
    pipeline, the k_cand roofline from HIP events around its launches, and the rows that fell back to the exact path."""
    import torch
    from anime_recommendations_amd import ops
    n = W.shape[0]
    q = torch.arange(nq, dtype=torch.int32, device="cuda")
    for _ in range(2):   # (twice: the caching allocator only reaches its steady state of blocks — workspace, two
        Wh = ops.rownorm(W)   # live output pairs — on the second call; a hipMalloc inside the timed region costs ms)
        idx, sim, nfb = ops.cosine_topk_mfma(Wh, q, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = {}
    for _ in range(reps):
        Wh = ops.rownorm(W)
        idx, sim, nfb = ops.cosine_topk_mfma(Wh, q, k, stats=stats)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    flops = 2.0 * nq * n * 128
    # roofline leg: HIP events around the k_cand launches (on the stream they run on) of one extra call of the whole
    # op — every query batch, with the threshold prior the later batches get from the first one; the timed call runs
    # its batches on ONE chain so that every timed launch has the chip to itself (the pipeline above runs two)
    acc = {}
    ops.cosine_topk_mfma(Wh, q, k, cand_timing=acc)
    cand_ms, cand_launches = acc.get("ms", 0.0), acc.get("launches", 0)
    tfk = flops / (cand_ms * 1e-3) / 1e12
    # all-pairs job (every key row a query row): a batch computes its dot products with the rows of later batches once
    # for both sides and skips the key tiles of earlier batches — the share of the 2 nq n D flops the MFMA kernel executes
    allpairs = bool(stats.get("allpairs", False))
    st = stats.get("starts", [0, nq])
    work = 1.0
    if allpairs:
        work = sum((st[b + 1] - st[b]) * (n if b == 0 else (st[1] + n - st[b])) for b in range(len(st) - 1)) / float(nq) / n
    # HBM bytes of all k_cand launches of ONE call of the profiled run (scripts/time_topk.py makes 3 calls of the same
    # job: n keys, nq queries, k)
    traffic = None
    if traffic_name:       # both instantiations (256- / 128-row workgroups: the learning batch and small jobs use the latter)
        parts = [pmc_traffic(traffic_name[0], [kern], source=["anirec_topk_mfma.hip", "ops.py"], calls=3)
                 for kern in ("k_cand<0, 8, false, false>", "k_cand<0, 4, false, false>", "k_cand<0, 8, false, true>",
                              "k_cand<0, 4, false, true>")]
        if any(p is not None for p in parts):
            traffic = sum(p for p in parts if p is not None)
    if traffic is not None and nq != traffic_name[1]:
        traffic = None      # only a profile of THIS job size is quoted (no scaling of a slice)
    rec = {"value": nq / dt, "unit": "queries/s", "ms": dt * 1e3, "k": k, "n_keys": n, "n_queries": nq,
           "fallback_rows": int(nfb), "rerun_rows": int(stats.get("rerun_rows", 0)),
           "batches": int(stats.get("batches", 0)), "learn_batches": int(stats.get("learn_batches", 0)),
           "chains": int(stats.get("lanes", 1)), "allpairs_shortcut": allpairs, "pipeline_tflops": flops / dt / 1e12,
           "roofline": {"kernel": "k_cand (v_mfma_f32_16x16x32_f16 scores + fused candidate filter), "
                                  "%d launches summed" % cand_launches,
                        "bound": "mfma", "achieved": tfk, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": tfk / MFMA_F16_PEAK_TFLOPS, "kernel_ms": cand_ms,
                        "pipeline_frac": flops / dt / 1e12 / MFMA_F16_PEAK_TFLOPS, "traffic": traffic}}
    if allpairs:
        # `achieved` / `frac` are what the matrix cores EXECUTED: the all-pairs job computes cosine(i, j) once for both
        # rows across query batches, so the MFMA kernel runs executed_flops_share of SURVEY 8(d)'s 2 nq n D flops
        # ("symmetry not exploited"); the figure on those algorithmic flops is kept under its own name
        rec["roofline"].update({
            "achieved": tfk * work, "frac": tfk * work / MFMA_F16_PEAK_TFLOPS, "executed_flops_share": work,
            "algorithmic_equivalent_tflops": tfk, "algorithmic_equivalent_frac": tfk / MFMA_F16_PEAK_TFLOPS,
            "note": "achieved / frac: the flops the MFMA kernel executed (executed_flops_share of 2 nq n D: every dot "
                    "product across query batches is computed once for both rows) over the summed k_cand time; "
                    "algorithmic_equivalent_* and pipeline_frac are the same times on the full 2 nq n D of SURVEY 8(d).  "
                    "Lists are bit-identical to the exact path and to the plain job (ANIREC_TOPK_SYM=0), which the "
                    "tests compare row for row"})
    if slice_note:
        rec["note"] = slice_note
    if cpu_baseline:
        from oracle import c_oracle
        Whn = Wh.cpu().numpy()
        nqc = 256 if n < 50_000 else 32
        t0 = time.perf_counter()
        c_oracle.cosine_topk(Whn, np.arange(nqc, dtype=np.int32), k)
        dtc = time.perf_counter() - t0
        rec["cpu_baseline"] = {"value": nqc / dtc, "unit": "queries/s", "cores": c_oracle.max_threads(),
                               "kind": "port", "sample": "%d queries, plain-C dot + top-%d per query" % (nqc, k)}
    del Wh, q
    return rec


def run_cosine_topk(cpu_baseline=True, trained=None):
    """BASELINE.json configs[3] as written: row-normalise + ALL-PAIRS cosine + top-100 over the 18 k x 128 anime and
    the 350 k x 128 user matrices (every row a query) — the headline legs.  k = 10, the reference's configured
    neighbour count (config/config.yaml:105 id_query_number), is reported beside them.  Embeddings: N(0, 0.05^2)
     and, when `trained` = (U, A) is given, the tables after one epoch of training on S7M."""
    import torch
    out = {}
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    Wu = torch.randn(350_000, 128, generator=g, device="cuda") * 0.05
    Wa = torch.randn(18_000, 128, generator=g, device="cuda") * 0.05
    out["anime_18k_allpairs_top100"] = _cosine_leg(Wa, 18_000, 100, cpu_baseline,
                                                   traffic_name=("cosine_topk_18k_k100", 18_000))
    out["users_350k_allpairs_top100"] = _cosine_leg(Wu, 350_000, 100, cpu_baseline, reps=2,
                                                    traffic_name=("cosine_topk_allpairs_k100", 350_000))
    out["anime_18k_allpairs_top10"] = _cosine_leg(Wa, 18_000, 10, False)
    out["users_350k_allpairs_top10"] = _cosine_leg(Wu, 350_000, 10, False, reps=2,
                                                   traffic_name=("cosine_topk_allpairs_k10", 350_000))
    out["users_350k_keys_65536q_top100"] = _cosine_leg(Wu, 65_536, 100, False, traffic_name=("cosine_topk_k100", 65_536),
                                                       slice_note="one 65 536-query slice of the all-pairs job")
    del Wu, Wa
    torch.cuda.empty_cache()
    if trained is not None:
        U, A = (torch.as_tensor(x, device="cuda") for x in trained)
        out["trained_s7m_anime_%d_allpairs_top100" % A.shape[0]] = _cosine_leg(A, A.shape[0], 100, False)
        out["trained_s7m_users_%d_allpairs_top100" % U.shape[0]] = _cosine_leg(U, U.shape[0], 100, False)
        del U, A
        torch.cuda.empty_cache()
    return out


def run_s7m_epoch(use_graph=True):
    """BASELINE.json configs[1] end to end: a 7 M-row user_stats.parquet-schema file (data.synth_user_stats,
    SURVEY §8(d) S7M: 15 000 users x 17 560 anime) read back from disk, id-encoded on the GPU
    (ingest.load_user_stats = get_df, neural_network.py:25-63) and trained for ONE epoch by trainer.fit
    (model.fit, :210-217) with the reference's hyper-parameters.  Returns the record and the trained tables."""
    import tempfile
    import torch
    from anime_recommendations_amd import data, ingest, trainer
    t0 = time.perf_counter()
    df = data.synth_user_stats(n_users=15_000, n_anime=17_560, n_ratings=7_000_000)
    t_gen = time.perf_counter() - t0
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "user_stats.parquet")
        df.to_parquet(path, index=False)
        n_rows = len(df)
        del df
        t0 = time.perf_counter()
        table = ingest.load_user_stats(path)
        torch.cuda.synchronize()
        t_load = time.perf_counter() - t0
    cfg = trainer.FitConfig(epochs=1, verbose=0, use_graph=use_graph)
    trainer.fit(table, trainer.FitConfig(epochs=1, verbose=0, use_graph=use_graph, test_size=len(table) - 20_000))  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = trainer.fit(table, cfg)
    torch.cuda.synchronize()
    t_fit = time.perf_counter() - t0
    n_train = len(table) - cfg.test_size
    steps = (n_train + cfg.batch_size - 1) // cfg.batch_size
    loop = res.step_loop_seconds[0]
    # an epoch once the engine exists (fit() builds it, captures the step graph, draws the initial weights on the host
    # and copies the tables and Adam slots back once per call): epochs 2 and 3 of a 3-epoch fit, timed inside fit
    # around shuffle + steps + metrics + hold-out validation + best-weights snapshot
    res3 = trainer.fit(table, trainer.FitConfig(epochs=3, verbose=0, use_graph=use_graph, patience=10))
    later = sorted(res3.epoch_seconds[1:])
    t_epoch = later[len(later) // 2] if later else float("nan")
    rec = {"value": n_train / t_fit, "unit": "ratings/s", "epoch_s": t_fit, "steps": steps,
           "later_epoch_s": t_epoch, "later_epoch_ratings_per_s": n_train / t_epoch,
           "ms_per_step": t_fit / steps * 1e3,
           # the step loop alone (engine.run over the %d steps, synchronised both sides); the rest of epoch_s is the
           # epoch shuffle, schedule upload, hold-out evaluation, best-weights snapshot and the copy of both tables
           # and the Adam slots back to the host
           "step_loop_s": loop, "step_loop_ms_per_step": loop / steps * 1e3, "step_loop_ratings_per_s": n_train / loop,
           "rows": n_rows, "n_users": table.n_users, "n_anime": table.n_anime,
           "loss": res.history["loss"][0], "val_loss": res.history["val_loss"][0],
           "load_user_stats_s": t_load, "synth_generation_s": t_gen,
           "what": "one epoch of trainer.fit (shuffle, %d steps of 10 000, hold-out validation, best-weights snapshot, "
                   "weights copied back to the host) on a 7 M-row synthetic user_stats.parquet read through "
                   "ingest.load_user_stats" % steps}
    return rec, (res.U, res.A)


def run_s109m_epoch(use_graph=True):
    """BASELINE.json configs[2] at N = 1 END TO END: 109 M synthetic ratings (350 000 users x 18 000 anime, SURVEY §8(d)
    S109M) generated in HBM with raw sparse ids and grouped by user like the real animelist, id-encoded and shuffled by
    ingest.encode_columns (= get_df, neural_network.py:25-63: Series.unique() positions, df.sample(frac=1,
    random_state=42)) and trained by trainer.fit (model.fit, :210-217) with the reference's hyper-parameters: 10 899
    steps of 10 000 per epoch, epoch shuffle, hold-out validation, best-weights snapshot.  Two epochs: the first pays
    for the engine, the graph capture and the host-side weight init; `later_epoch_s` is the second."""
    import torch
    from anime_recommendations_amd import ingest, trainer
    n_users, n_anime, n = 350_000, 18_000, 109_000_000
    dev = torch.device("cuda")
    t0 = time.perf_counter()
    ui, ai, t = synth_ratings(n_users, n_anime, n, dev)
    order = torch.sort(ui, stable=True)[1]                      # the raw table is grouped by user
    cols = {"user_id": ui[order] * 3 + 7, "anime_id": ai[order] * 2 + 1, "rating": t[order].double()}
    del ui, ai, t, order
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    table = ingest.encode_columns(cols)
    torch.cuda.synchronize()
    t_enc = time.perf_counter() - t0
    del cols
    cfg = trainer.FitConfig(epochs=2, verbose=0, use_graph=use_graph, patience=10)
    t0 = time.perf_counter()
    res = trainer.fit(table, cfg)
    torch.cuda.synchronize()
    t_fit = time.perf_counter() - t0
    n_train = len(table) - cfg.test_size
    steps = (n_train + cfg.batch_size - 1) // cfg.batch_size
    later = res.epoch_seconds[1]
    rec = {"value": n_train / later, "unit": "ratings/s", "later_epoch_s": later, "first_epoch_s": res.epoch_seconds[0],
           "fit_2_epochs_s": t_fit, "steps_per_epoch": steps, "step_loop_s": res.step_loop_seconds[1],
           "step_loop_ms_per_step": res.step_loop_seconds[1] / steps * 1e3,
           "step_loop_ratings_per_s": n_train / res.step_loop_seconds[1],
           "rows": n, "n_users": table.n_users, "n_anime": table.n_anime,
           "loss": res.history["loss"], "val_loss": res.history["val_loss"],
           "encode_and_shuffle_s": t_enc, "synth_generation_s": t_gen,
           "what": "two epochs of trainer.fit on 109 M synthetic ratings resident in HBM (ingest.encode_columns -> "
                   "trainer.fit): %d steps of 10 000 per epoch, epoch shuffle, hold-out validation, best-weights "
                   "snapshot; value = the second epoch" % steps}
    del table, res
    torch.cuda.empty_cache()
    return rec


ING_KERNELS = {"k_ing_init": 1, "k_enc_init": 2, "k_ing_span": 1, "k_ing_front<true>": 1, "k_nl_clear": 1, "k_nl_insert": 1, "k_nl_count": 1,
               "k_nl_filter<true>": 1, "k_scan_spine": 3, "k_ing_compact": 1, "k_enc_first": 1, "k_enc_first_lds": 1,
               "k_enc_bits": 2, "k_bits_reduce": 2, "k_bits_apply": 2, "k_enc_rank": 2, "k_enc_emit": 1,
               "k_enc_emit_lds": 1}


def run_ingest(cpu_baseline=True):
    """the preprocess step + id encoding on 109 M raw rows resident in HBM
    (grouped by user like the real animelist; ~0.5 % duplicate rows, plan-to-watch rows dropped,
    users with < 250 surviving ratings dropped), timed next to the pandas restatement of
    preprocess.py on a bounded sample of the same table."""
    import torch
    from anime_recommendations_amd import ingest
    n, n_users, n_anime = 109_000_000, 350_000, 18_000
    g = torch.Generator(device="cuda")
    g.manual_seed(17)
    cols = {
        "user_id": torch.sort(torch.randint(0, n_users, (n,), generator=g, device="cuda", dtype=torch.int32))[0],
        "anime_id": torch.randint(0, n_anime, (n,), generator=g, device="cuda", dtype=torch.int32),
        "rating": torch.randint(0, 11, (n,), generator=g, device="cuda", dtype=torch.int32).double(),
        "watching_status": torch.randint(1, 7, (n,), generator=g, device="cuda", dtype=torch.int32),
        "watched_episodes": torch.randint(0, 26, (n,), generator=g, device="cuda", dtype=torch.int32),
    }
    dup_dst = torch.unique(torch.randint(0, n, (n // 200,), generator=g, device="cuda"))
    dup_src = torch.clamp(dup_dst - torch.randint(1, 50, (dup_dst.numel(),), generator=g, device="cuda"), min=0)
    for k in cols:                                       # duplicate rows near their originals (same user)
        cols[k][dup_dst] = cols[k].clone()[dup_src]      # (unique targets, untouched sources: reproducible)

    def once():
        out = ingest.preprocess_columns(cols, num_reviews=250, drop_plan=True)
        enc_u = ingest.encode_ids(out["user_id"], out.bounds["user_id"])
        enc_a = ingest.encode_ids(out["anime_id"], out.bounds["anime_id"])
        return out, enc_u, enc_a
    once()
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        out, enc_u, enc_a = once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    m = int(out["user_id"].numel())
    alg_bytes = 24 * n + (24 + 8) * m        # read 5 columns; write 5 columns + 2 index columns of the survivors
    gbs = alg_bytes / dt / 1e9
    rec = {"value": n / dt, "unit": "rows/s", "ms": dt * 1e3, "rows_in": n, "rows_out": m,
           "n_users": int(enc_u[1].numel()), "n_anime": int(enc_a[1].numel()),
           "roofline": {"kernel": "ingest pipeline (27 launches; k_ing_front — row filters, LDS dedupe and per-user "
                                  "counts of an 8 192-row chunk and the rows its last user reaches past it — and "
                                  "k_ing_compact are 3/4 of it)",
                        "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes": alg_bytes,
                        "limiter": "k_ing_front is bound by instruction issue (PMC, profiles/r04_pmc_ingest_front.txt: "
                                   "3 066 vector + 2 957 scalar instructions per wave, the four waves of a SIMD "
                                   "together in execution 106 % of the time), not by HBM",
                        "traffic": pmc_traffic("ingest", dict(ING_KERNELS, k_ing_id_max=1),
                                               source=["anirec_ingest.hip", "ingest.py"]),
                        # the id-bounds pass (both id columns read once more: 0.87 GB) is only run for columns that
                        # come without bounds; rounds 1-3 did it with two torch reductions the kernel list never saw
                        "traffic_without_bounds_pass": pmc_traffic("ingest", ING_KERNELS,
                                                                   source=["anirec_ingest.hip", "ingest.py"])}}
    if cpu_baseline:
        import pandas as pd
        from oracle import ingest_oracle
        ns = 4_000_000
        df = pd.DataFrame({k: v[:ns].cpu().numpy() for k, v in cols.items()})
        t0 = time.perf_counter()
        pre = ingest_oracle.preprocess(df, 250, drop_plan=True)
        ingest_oracle.encode(pre["user_id"])
        ingest_oracle.encode(pre["anime_id"])
        dtc = time.perf_counter() - t0
        rec["cpu_baseline"] = {"value": ns / dtc, "unit": "rows/s", "cores": 1, "kind": "port",
                               "sample": "first %d rows: pandas restatement of preprocess.py drop_useless + "
                                         "scale_ratings + Series.unique() encoding (pandas is single-threaded)" % ns}
    del cols, out
    torch.cuda.empty_cache()
    return rec


def run_predict_topk(cpu_baseline=True):
    """BASELINE.json configs[4] as model_recs consumes it: per user the top-10 unwatched anime by predicted
    rating (100 k users x 18 k anime, ~25 % watched) — the rating grid is never written."""
    import torch
    from anime_recommendations_amd import ops
    n_u, n_a, nq, k = 350_000, 18_000, 100_000, 10
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    U = torch.randn(n_u, 128, generator=g, device="cuda") * 0.05
    A = torch.randn(n_a, 128, generator=g, device="cuda") * 0.05
    head = dict(w=1.3, b=0.1, gamma=0.9, beta=-0.2, mov_mean=0.05, mov_var=0.4)
    users = torch.arange(nq, dtype=torch.int32, device="cuda")
    shape = (nq, (n_a + 31) // 32)
    watched = torch.randint(-2 ** 31, 2 ** 31 - 1, shape, generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
    watched &= torch.randint(-2 ** 31, 2 ** 31 - 1, shape, generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
    ops.predict_topk_mfma(U, A, head, users, k, watched)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        idx, p, nfb = ops.predict_topk_mfma(U, A, head, users, k, watched)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ops.topk_mfma_timing(True)
    ops.predict_topk_mfma(U, A, head, users, k, watched)
    cand_ms, cand_launches = ops.topk_mfma_timing(False)
    flops = 2.0 * nq * n_a * 128
    tfk = flops / (cand_ms * 1e-3) / 1e12
    t0 = time.perf_counter()
    ops.predict_topk(U, A, head, users, k, watched)
    torch.cuda.synchronize()
    dte = time.perf_counter() - t0
    rec = {"value": nq / dt, "unit": "users/s", "ms": dt * 1e3, "k": k, "fallback_rows": int(nfb),
           "ratings_per_s": nq * n_a / dt, "exact_fp32_path_ms": dte * 1e3,
           "roofline": {"kernel": "k_cand<masked> (f16 MFMA cosine + watched mask + candidate filter), %d launches; "
                                  "18 k keys = 141 tiles only: ~60 appends per tile-wave, append-bound" % cand_launches,
                        "bound": "mfma", "achieved": tfk, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": tfk / MFMA_F16_PEAK_TFLOPS, "kernel_ms": cand_ms,
                        # scripts/time_predict_topk.py calls the op twice: all k_cand launches of ONE call
                        "traffic": pmc_traffic("ptk", ["k_cand<0, 8, true, false>"], source="anirec_topk_mfma.hip", calls=2)}}
    if cpu_baseline:
        from oracle import c_oracle
        nc = 256
        t0 = time.perf_counter()
        G = c_oracle.predict_grid(U[:nc].cpu().numpy(), A.cpu().numpy(), dict(head, m=[0] * 4, v=[0] * 4),
                                  np.arange(nc, dtype=np.int32))
        for j in range(nc):
            np.argpartition(-G[j], k)[:k]
        dtc = time.perf_counter() - t0
        rec["cpu_baseline"] = {"value": nc / dtc, "unit": "users/s", "cores": c_oracle.max_threads(), "kind": "port",
                               "sample": "%d users x 18000 anime: plain-C model.predict + NumPy argpartition" % nc}
    del U, A, watched
    torch.cuda.empty_cache()
    return rec


def run_user_recs(cpu_baseline=True):
    """favourites of every user (80th percentile of their own ratings) from 109 M
    ratings, then for 65 536 query users the 10 anime their 10 most similar users favourited most."""
    import torch
    from anime_recommendations_amd import ops, recs
    n_users, n_anime, n = 350_000, 18_000, 109_000_000
    dev = torch.device("cuda")
    ui, ai, t = synth_ratings(n_users, n_anime, n, dev)
    order = torch.sort(ui, stable=True)[1]            # the rating table in its raw order: grouped by user
    ui, ai, t = ui[order], ai[order], t[order]
    del order
    r = t.double()
    for _ in range(3):          # the two 0.8 GB result / workspace blocks settle in torch's caching allocator
        fav, thr = recs.user_favourites(ui, ai, r, n_users, n_anime)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        fav, thr = recs.user_favourites(ui, ai, r, n_users, n_anime)
    torch.cuda.synchronize()
    dt_f = (time.perf_counter() - t0) / reps
    nq, k_sim, n_recs = 65_536, 10, 10
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    U = torch.randn(n_users, 128, generator=g, device="cuda") * 0.05
    q = torch.arange(nq, dtype=torch.int32, device="cuda")
    sim, _, _ = ops.cosine_topk_mfma(ops.rownorm(U), q, k_sim)
    recs.user_recs(fav, n_anime, q, sim, n_recs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out_a, out_c = recs.user_recs(fav, n_anime, q, sim, n_recs)
    torch.cuda.synchronize()
    dt_r = (time.perf_counter() - t0) / reps
    ww = (n_anime + 31) // 32
    fav_bytes = 16 * n + (n_users * ww * 4)               # the three columns read once, the bit rows written once
    rec_bytes = nq * (k_sim + 1) * ww * 4                 # the similar users' and the query's bit rows
    rec = {"value": nq / dt_r, "unit": "queries/s", "ms": dt_r * 1e3,
           "favourites": {"ms": dt_f * 1e3, "ratings_per_s": n / dt_f,
                          "roofline": {"kernel": "k_rec_count, 3-launch scan, k_rec_percentile with the favourite-bit rows "
                                                 "built in LDS (table grouped by user); 8 launches", "bound": "hbm",
                                       "achieved": fav_bytes / dt_f / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": fav_bytes / dt_f / 1e9 / HBM_PEAK_GBS,
                                       "traffic": pmc_traffic("recs", {"k_rec_count": 1, "k_rec_scan_reduce": 1,
                                                                       "k_rec_scan_spine": 1, "k_rec_scan_apply": 1,
                                                                       "k_rec_clear": 1, "k_rec_scatter": 1,
                                                                       "k_rec_percentile": 1, "k_rec_favbits": 1},
                                                              source="anirec_recs.hip")}},
           "roofline": {"kernel": "k_user_recs<3, 4> (one workgroup per query: 11 bit rows of 2.25 KB -> bit-sliced counts in registers, block-wide binary searches for the cut -> top-10)",
                        "bound": "hbm", "achieved": rec_bytes / dt_r / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": rec_bytes / dt_r / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic("recs", "k_user_recs<3, 4>", source="anirec_recs.hip")}}
    if cpu_baseline:
        from oracle import recs_oracle
        nu_s = 2_000                                       # users 0..1999 of the same table
        m = (ui < nu_s)
        us, as_, rs = ui[m].cpu().numpy(), ai[m].cpu().numpy(), r[m].cpu().numpy()
        t0 = time.perf_counter()
        _, fav_o = recs_oracle.favourites(us, as_, rs, nu_s)
        dtf = time.perf_counter() - t0
        rng = np.random.default_rng(0)
        t0 = time.perf_counter()
        for qq in range(200):
            recs_oracle.value_counts_of_similar_favourites(fav_o, qq, rng.integers(0, nu_s, k_sim).tolist())
        dtr = time.perf_counter() - t0
        rec["cpu_baseline"] = {"value": 200 / dtr, "unit": "queries/s", "cores": 1, "kind": "port",
                               "favourites_users_per_s": nu_s / dtf,
                               "sample": "%d users: np.percentile + filter per user; 200 queries: pandas ravel + "
                                         "value_counts (user_recs.py:377-404, :732-745)" % nu_s}
    del ui, ai, t, r, U, fav
    torch.cuda.empty_cache()
    return rec


def run_gather_roofline():
    """The embedding-forward kernel body (two 512-B row gathers + three dot-128 reductions per rating,
    k_predict_pairs == k_fwd without the batch bookkeeping) on 4 M random pairs: the HBM gather rate
    the kernel reaches when a launch is not 10 000 ratings (10 MB, latency-bound) long."""
    import torch
    from anime_recommendations_amd import ops
    n_u, n_a, n = 350_000, 18_000, 4_000_000
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    U = torch.randn(n_u, 128, generator=g, device="cuda") * 0.05
    A = torch.randn(n_a, 128, generator=g, device="cuda") * 0.05
    ui = torch.randint(0, n_u, (n,), generator=g, device="cuda", dtype=torch.int32)
    ai = torch.randint(0, n_a, (n,), generator=g, device="cuda", dtype=torch.int32)
    head = dict(w=1.3, b=0.1, gamma=0.9, beta=-0.2, mov_mean=0.05, mov_var=0.4)
    ops.predict_pairs(U, A, head, ui, ai)
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        ops.predict_pairs(U, A, head, ui, ai)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    gbs = n * 1036 / dt / 1e9
    return {"value": n / dt, "unit": "pairs/s", "ms": dt * 1e3,
            "roofline": {"kernel": "k_predict_pairs (embedding forward body), 4 M pairs", "bound": "hbm",
                         "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "traffic": None}}


def run_predict_grid(cpu_baseline=True):
    """BASELINE.json configs[4]: predicted ratings of all 18 000 anime for 100 000 query users
    (the full fp32 grid is written: 7.2 GB, HBM-write-bound)."""
    import torch
    from anime_recommendations_amd import ops
    n_u, n_a, nq = 350_000, 18_000, 100_000
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    U = torch.randn(n_u, 128, generator=g, device="cuda") * 0.05
    A = torch.randn(n_a, 128, generator=g, device="cuda") * 0.05
    head = dict(w=1.3, b=0.1, gamma=0.9, beta=-0.2, mov_mean=0.05, mov_var=0.4)
    users = torch.arange(nq, dtype=torch.int32, device="cuda")
    out = torch.empty(nq, n_a, dtype=torch.float32, device="cuda")
    ops.predict_grid_mfma(U, A, head, users, out=out)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        ops.predict_grid_mfma(U, A, head, users, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    gbs = nq * n_a * 4 / dt / 1e9
    rec = {"value": nq * n_a / dt, "unit": "ratings/s", "ms": dt * 1e3,
           "roofline": {"kernel": "k_predict_mfma2 (split-f16 MFMA, anime as the row operand: 16-B non-temporal row-quad "
                                  "stores, sigmoid + stores of a tile under the MFMAs of the next)", "bound": "hbm",
                        "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "traffic": pmc_traffic("pgrid", "k_predict_mfma2<0>", source="anirec_predict_mfma.hip")}}
    if cpu_baseline:
        from oracle import c_oracle
        Un, An = U[:4096].cpu().numpy(), A.cpu().numpy()
        nqc = 256
        t0 = time.perf_counter()
        c_oracle.predict_grid(Un, An, dict(head, m=[0] * 4, v=[0] * 4), np.arange(nqc, dtype=np.int32))
        dtc = time.perf_counter() - t0
        rec["cpu_baseline"] = {"value": nqc * n_a / dtc, "unit": "ratings/s", "cores": c_oracle.max_threads(),
                               "kind": "port", "sample": "%d users x %d anime, plain-C restatement of model.predict" % (nqc, n_a)}
    del U, A, out
    torch.cuda.empty_cache()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--batch", type=int, default=10_000, help="ratings per GPU per step")
    ap.add_argument("--workload", default="s109m", choices=sorted(WORKLOADS))
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary workloads")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 or args.gpus > 1 or os.environ.get("ANIREC_FORCE_DIST") == "1":
        from anime_recommendations_amd import dist_bench
        return dist_bench.main(args)

    res = run_single(args.workload, args.steps, args.warmup, args.batch, use_graph=not args.no_graph,
                     cpu_baseline=not args.no_cpu_baseline)
    n_users, n_anime = WORKLOADS[args.workload]
    line = {
        "metric": "training_ratings_per_sec", "value": res["value"], "unit": "ratings/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "%s: train step on %d users x %d anime tables, D=128, batch %d/GPU, "
                               "L2 1e-4, Keras-2.12 Adam, lr=lrfn(0)=1e-5" % (args.workload, n_users, n_anime, args.batch),
                   "global_batch": args.batch, "parallelism": "dp1"},
        "roofline": res["roofline"],
        "step_roofline": res["step_roofline"],
        "cpu_baseline": res.get("cpu_baseline"),
        "adam": res["adam"],
        "kernels_ms": res["kernels_ms"],
        "embed_fwd_GBps": res["fwd_gbs"], "embed_bwd_GBps": res["bwd_gbs"],
        "final_loss": res["loss"],
    }
    for k in ("lazy_kernels_ms", "dense_kernel_roofline"):
        if k in res:
            line[k] = res[k]
    if not args.no_also:
        other = "s7m" if args.workload == "s109m" else "s109m"
        r2 = run_single(other, args.steps, args.warmup, args.batch, use_graph=not args.no_graph,
                        cpu_baseline=not args.no_cpu_baseline)
        line["also"] = {}
        shape_rec = {"value": r2["value"], "unit": "ratings/s", "ms_per_step": r2["ms_per_step"], "adam": r2["adam"],
                     "roofline": r2["roofline"], "step_roofline": r2["step_roofline"], "kernels_ms": r2["kernels_ms"],
                     "cpu_baseline": r2.get("cpu_baseline"),
                     "what": "%d steps on the %s table shape, i.i.d. synthetic batches resident in HBM" % (args.steps, other)}
        trained = None
        if other == "s7m":
            # configs[1]: the real thing first (a 7 M-row file through get_df -> model.fit), the table-shape
            # micro-benchmark (same kernels, no epoch bookkeeping) beside it
            epoch_rec, trained = run_s7m_epoch(use_graph=not args.no_graph)
            epoch_rec["table_shape_microbench"] = shape_rec
            line["also"]["s7m"] = epoch_rec
        else:
            line["also"][other] = shape_rec
        if args.workload == "s109m":
            line["also"]["s109m_epoch"] = run_s109m_epoch(use_graph=not args.no_graph)
        line["also"]["cosine_topk"] = run_cosine_topk(cpu_baseline=not args.no_cpu_baseline, trained=trained)
        from anime_recommendations_amd import ops as _ops
        _ops.release_workspaces()      # (the all-pairs job keeps its multi-GB workspace between calls)
        line["also"]["predict_grid_100k_x_18k"] = run_predict_grid(cpu_baseline=not args.no_cpu_baseline)
        line["also"]["predict_topk_100k_users_x_18k"] = run_predict_topk(cpu_baseline=not args.no_cpu_baseline)
        line["also"]["embed_fwd_gather_4M_pairs"] = run_gather_roofline()
        line["also"]["ingest_109m_rows"] = run_ingest(cpu_baseline=not args.no_cpu_baseline)
        line["also"]["user_recs_65536_queries"] = run_user_recs(cpu_baseline=not args.no_cpu_baseline)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
