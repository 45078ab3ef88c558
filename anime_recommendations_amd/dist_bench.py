"""bench.py's N>1 leg: one rank per MI355X, RCCL over xGMI.  Weak scaling: every rank takes ``--batch``
ratings of each global batch (global batch = N x batch).

Launch: either ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N`` (the
ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or plainly ``python bench.py --gpus N``:
the parent then starts its N ranks itself as child processes — before it has made any GPU call — and waits.

The headline number is the user-sharded mode (dist.py).  After it, on the same ranks and the same synthetic
ratings, the two replicated-table modes are timed for a few steps (``also.dp_modes``): the literal
"replicated variables + dense all-reduce of both tables" baseline of the reference's TPU branch
(neural_network.py:173-178) and its reduce-scatter -> shard-Adam -> all-gather form.  The ONE JSON line is printed at
the end; if the extras stall, a watchdog prints it without them (``also.extras_error``) and the process exits with
status 3, so the scaling number survives and the driver still sees that something went wrong.
"""
from __future__ import annotations

import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv):
    """Start the N ranks as fresh child processes (no GPU call has been made in this process: children are
    ordinary ``python bench.py ...`` processes with the torchrun environment variables set)."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, p.wait() or 0)
    return rc


def main(args):
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, [os.path.abspath(sys.argv[0])] + sys.argv[1:]))
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # ANIREC_DIST_BACKEND=gloo lets several ranks share one GPU for a functional rehearsal
    backend = os.environ.get("ANIREC_DIST_BACKEND", "nccl")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    _run(args, rank, world, dev)
    dist.barrier()
    dist.destroy_process_group()


def _sharded_topk_leg(rank, world, dev, k=100):
    """The second half of BASELINE.json's metric at N GPUs: the whole similar-users job — cosine top-k of
    all 350 000 users against all 350 000 keys, k = 100 (configs[3]) — with the query rows sharded over the ranks
    (independent units: no collective in the loop; each rank keeps its [nq/N, k] block).  Strong scaling of a
    fixed job.  Never fails or stalls the training bench: a rank that hits an error reports it, and the only
    collectives (one MAX, one MIN) are outside the guarded region so every rank always reaches them."""
    import torch
    import torch.distributed as dist
    n, nq = 350_000, 350_000
    my_ms, err, lo, hi = -1.0, None, 0, 0
    try:
        from . import ops
        from .dist_infer import shard_bounds
        g = torch.Generator(device=dev)
        g.manual_seed(7)
        W = torch.randn(n, 128, generator=g, device=dev) * 0.05
        Wh = ops.rownorm(W)
        lo, hi = shard_bounds(nq, rank, world)
        q = torch.arange(lo, hi, dtype=torch.int32, device=dev)
        ops.cosine_topk_mfma(Wh, q, k)
        torch.cuda.synchronize()
        reps = 2
        t0 = time.perf_counter()
        for _ in range(reps):
            ops.cosine_topk_mfma(Wh, q, k)      # batches of <= 65 536 queries inside
        torch.cuda.synchronize()
        my_ms = (time.perf_counter() - t0) / reps * 1e3
        del W, Wh
        torch.cuda.empty_cache()
    except Exception as exc:                                   # noqa: BLE001 - reported, never raised
        err = "%s: %s" % (type(exc).__name__, exc)
    stat = torch.tensor([my_ms, 0.0 if err else 1.0], dtype=torch.float64, device=dev)
    worst = stat.clone()
    dist.all_reduce(worst[0:1], op=dist.ReduceOp.MAX)
    dist.all_reduce(stat[1:2], op=dist.ReduceOp.MIN)
    if float(stat[1]) < 1.0:
        return {"error": err or "another rank failed"}
    ms = float(worst[0])
    return {"value": nq / (ms * 1e-3), "unit": "queries/s", "ms": ms, "k": k, "n_gpus": world,
            "queries_per_rank": hi - lo, "scaling": "strong"}


def _train_leg(mode, data, tables, n_users, n_anime, B, K, W, inst, rank, world, dev):
    """K timed steps of one data-parallel mode (barrier + synchronize on both sides, MAX over ranks), then an
    instrumented pass with HIP events on the engine's stream around each step half and collective."""
    import torch
    import torch.distributed as dist
    from .dist import DistTrainEngine
    from . import schedule
    ui, ai, t = data
    U, A = tables
    total = W + K + inst
    eng = DistTrainEngine(n_users, n_anime, B, device=dev, mode=mode)
    eng.set_head(w=1.2)
    eng.set_weights(U, A)
    n_used = total * B * world
    perm = torch.arange(n_used, device=dev)
    eng.set_epoch_global(ui[:n_used], ai[:n_used], t[:n_used], perm, schedule.adam_alphas(1e-5, 1, total))
    del perm
    if W:
        eng.run(W)
    eng.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    eng.run(K)
    t_issue = time.perf_counter() - t0           # host time to ENQUEUE the K steps (3 C calls + collectives each)
    eng.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0, t_issue], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt, t_issue = float(dt[0]), float(dt[1])
    rec = eng.read_state()
    assert int(rec["step_fwd"]) == W + K and np.isfinite(rec["last_loss"])

    e = eng.eng
    names = ("front", "gather", "mid", "reduce", "back", "allgather_w")
    evs = {k: [] for k in names}

    def timed(name, fn):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record(e.stream)
        fn()
        e1.record(e.stream)
        evs[name].append((e0, e1))

    done = 0
    if eng.loop:
        with torch.cuda.stream(e.stream):
            e.stepper_begin(W + K, inst)
            while done < inst:
                blk = min(e.arena_steps, inst - done)
                e.prep(W + K + done, blk)
                e.stepper_block(blk)
                for _ in range(blk):
                    timed("front", e.step_front)
                    timed("gather", eng._all_gather_packets)
                    timed("mid", e.step_mid)
                    timed("reduce", eng._reduce_dense)
                    timed("back", e.step_back)
                    if mode == "replicated_rs":
                        timed("allgather_w", eng._all_gather_rows)
                done += blk
    eng.synchronize()
    torch.cuda.synchronize()
    kern_ms = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in evs.items() if v}
    out = {"value": K * B * world / dt, "unit": "ratings/s", "ms_per_step": dt / K * 1e3, "steps": K,
           "host_issue_ms_per_step": t_issue / K * 1e3,
           # share of the step the host needs just to enqueue it: near 1.0 the Python loop, not the GPU, sets the pace
           "host_bound_frac": min(1.0, t_issue / dt), "stage_ms": kern_ms,
           "loop": "native (anirec_dist_run: RCCL from C)" if eng.native else "python (torch.distributed collectives)",
           "final_loss": float(rec["last_loss"]), "local_rows": int(e.rows),
           "user_row_adam": "lazy" if e.lazy else "dense",
           "adam_rows": int(e.adam_rows[1] - e.adam_rows[0]) if (e.adam_rows[0] | e.adam_rows[1]) else int(e.rows),
           "dense_grad_bytes": int(e.dense_grad.numel() * 4) if e.dense_grad is not None else 0}
    eng.close()
    del eng
    torch.cuda.empty_cache()
    return out


def _run(args, rank, world, dev):
    import torch
    import torch.distributed as dist
    import bench
    n_users, n_anime = bench.WORKLOADS[args.workload]
    B, K, W = args.batch, args.steps, args.warmup
    inst = min(K, 32)                              # instrumented per-stage pass
    # identical synthetic ratings on every rank (same seed, same device type); each rank keeps its share
    ui, ai, t = bench.synth_ratings(n_users, n_anime, (W + K + inst) * B * world, dev)
    U, A = bench.init_tables(n_users, n_anime, dev)
    mode = os.environ.get("ANIREC_DP_MODE", "sharded")
    head = _train_leg(mode, (ui, ai, t), (U, A), n_users, n_anime, B, K, W, inst, rank, world, dev)
    rows = head["adam_rows"] if mode != "sharded" else head["local_rows"]
    adam_bytes = bench.ADAM_BYTES_PER_ELEM * rows * 128
    gather_bytes = (bench.FWD_BYTES_PER_RATING + bench.BWD_BYTES_PER_RATING) * B
    dense_alg_bytes = adam_bytes + gather_bytes        # the dense update of every row this rank owns
    if head["user_row_adam"] == "lazy":
        # what the rank's step moves with lazy user rows: the gathers, the ~B touched user rows' W, M, V read and written
        # twice (catch-up, sparse step), 1/8 of a 24 B/element flush pass over its user rows, the dense 28 B/element
        # update of the replicated anime rows (an estimate from the launch shapes, not a counter)
        n_local = rows - n_anime
        step_bytes = (gather_bytes + 2 * min(B, n_local) * 128 * 4 * 3 * 2 + 24 * n_local * 128 / 8
                      + bench.ADAM_BYTES_PER_ELEM * n_anime * 128)
    else:
        step_bytes = dense_alg_bytes
    gbs = step_bytes / (head["ms_per_step"] * 1e-3) / 1e9
    line = {
        "metric": "training_ratings_per_sec", "value": head["value"], "unit": "ratings/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": head["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "%s: train step on %d users x %d anime tables, D=128, batch %d/GPU "
                               "(global %d), L2 1e-4, Keras-2.12 Adam, lr=lrfn(0)=1e-5"
                               % (args.workload, n_users, n_anime, B, B * world),
                   "global_batch": B * world,
                   "parallelism": {"sharded": "dp%d by user (mode `sharded`, the default — NOT north_star's literal "
                                              "replicated tables, which is also.dp_modes.replicated): user table + Adam state "
                                              "sharded with %s user-row Adam, anime table replicated with dense RCCL "
                                              "all-reduce, head packets all-gathered" % (world, head["user_row_adam"]),
                                   "replicated": "dp%d replicated tables, dense all-reduce of both tables" % world,
                                   "replicated_rs": "dp%d replicated tables, reduce-scatter -> shard Adam -> all-gather"
                                                    % world}[mode]},
        "roofline": {"kernel": "whole step of one rank (%s Adam over its %d rows + embedding fwd/bwd of its batch; the "
                               "collectives' wire time is inside the step time)" % (head["user_row_adam"], rows),
                     "bound": "hbm", "achieved": gbs, "peak": bench.HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": gbs / bench.HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": step_bytes, "avg_launch_ms": head["ms_per_step"],
                     "vs_dense_algorithm": {"algorithmic_bytes_per_step": dense_alg_bytes,
                                            "ratio_to_hbm_peak": dense_alg_bytes / (head["ms_per_step"] * 1e-3) / 1e9
                                            / bench.HBM_PEAK_GBS}},
        "cpu_baseline": None,
        "host_issue_ms_per_step": head["host_issue_ms_per_step"], "host_bound_frac": head["host_bound_frac"],
        "step_loop": head["loop"],
        "kernels_ms": head["stage_ms"],
        "final_loss": head["final_loss"],
        "also": {},
    }

    printed = threading.Event()

    def emit():
        if rank == 0 and not printed.is_set():
            printed.set()
            print(json.dumps(line), flush=True)

    # the extras below involve collectives of their own: if any rank stalls in them, every rank gives up after the
    # timeout: rank 0 prints the headline line with the error recorded, and every rank exits NON-ZERO (os._exit: the
    # other ranks may sit in a collective) so that the stall is visible to whoever launched the bench
    def give_up():
        line["also"]["extras_error"] = "timed out after %s s" % timeout_s
        emit()
        os._exit(3)

    timeout_s = float(os.environ.get("ANIREC_BENCH_EXTRAS_TIMEOUT", "300"))
    dog = None
    if not args.no_also:
        dog = threading.Timer(timeout_s, give_up)
        dog.daemon = True
        dog.start()
        try:
            Kx, Wx = min(K, 40), min(W, 8)
            modes = {}
            for m in ("sharded", "replicated", "replicated_rs"):
                if m == mode:
                    continue
                modes[m] = _train_leg(m, (ui, ai, t), (U, A), n_users, n_anime, B, Kx, Wx, min(inst, 8), rank, world, dev)
            modes[mode] = {k: head[k] for k in ("value", "unit", "ms_per_step", "steps", "host_bound_frac", "stage_ms",
                                                "adam_rows", "dense_grad_bytes", "user_row_adam")}
            line["also"]["dp_modes"] = modes
        except Exception as exc:                               # noqa: BLE001 - reported; the watchdog covers hangs
            line["also"]["dp_modes"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        del ui, ai, t, U, A
        torch.cuda.empty_cache()
        line["also"]["cosine_topk_users_350k_allpairs_top100"] = _sharded_topk_leg(rank, world, dev)
        dog.cancel()
    emit()
