"""GPU tests of the multi-GPU path with the REAL HIP step halves.
  * two ranks share cuda:0 and talk over gloo, in each of the three data-parallel modes: the result must match the
    oracle stepping on the global batches and the replicated tables must stay bit-identical across ranks;
  * backend "nccl" (= RCCL), world size 1, ANIREC_DIST_LOOP=1: the N>1 loop — all_gather_into_tensor, all_reduce /
    reduce_scatter_tensor on the engine's stream, the forked user-row Adam — runs on hardware and must equal the
    single-engine run bit for bit;
  * two ranks on two GPUs over RCCL when the box has two (skipped on a one-GPU box; the driver's 8-GPU node runs it)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import anirec_oracle as orc

pytestmark = pytest.mark.gpu


def _problem():
    rng = np.random.default_rng(21)
    n_u, n_a, n = 3001, 700, 11 * 2000 - 333       # 11 global batches: a full lazy window of 8, then a ragged one
    U = rng.uniform(-0.05, 0.05, (n_u, 128)).astype(np.float32)
    A = rng.uniform(-0.05, 0.05, (n_a, 128)).astype(np.float32)
    ui = rng.integers(0, n_u, n)
    ai = (rng.zipf(1.15, n) - 1) % n_a
    t = (rng.integers(0, 11, n) / 10).astype(np.float32)
    return U, A, ui, ai, t, rng.permutation(n)


def _worker(rank, world, port, out_dir, mode="sharded", backend="gloo", lazy=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda:%d" % (rank if backend == "nccl" else 0))
    if backend == "nccl":
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from anime_recommendations_amd import schedule
        from anime_recommendations_amd.dist import DistTrainEngine
        U, A, ui, ai, t, perm = _problem()
        eng = DistTrainEngine(U.shape[0], A.shape[0], 1000, l2=1e-4, arena_steps=4, device=dev, mode=mode, lazy=lazy)
        assert eng.eng.lazy == bool(lazy)
        eng.set_head(w=1.2)
        eng.set_weights(U, A)
        tu, ta, tt, tp = (torch.from_numpy(np.asarray(x)).to(dev) for x in (ui, ai, t, perm))
        n_steps = (len(perm) + 1999) // 2000
        eng.set_epoch_global(tu, ta, tt, tp, schedule.adam_alphas(3e-5, 1, n_steps))
        eng.reset_metrics()
        eng.run(n_steps)
        loss, mse = eng.epoch_metrics()
        vl, vm = eng.evaluate(tu[:500], ta[:500], tt[:500])
        Ufull = eng.U.cpu().numpy()
        Aloc = eng.A.cpu()
        opt = eng.optimizer_state(iterations=n_steps)       # collective: the full-table Adam slots (model.save)
        for tbl in ([eng.A] if mode == "sharded" else [eng.A, eng.eng.U]):
            tbl = tbl.contiguous() if backend == "nccl" else tbl.cpu()
            a_all = [torch.empty_like(tbl) for _ in range(world)]
            dist.all_gather(a_all, tbl)
            assert all((x == a_all[0]).all() for x in a_all)
        if rank == 0:
            rec = eng.read_state()
            np.savez(os.path.join(out_dir, "dist.npz"), U=Ufull, A=Aloc.numpy(), loss=loss, mse=mse, vl=vl, vm=vm,
                     w=rec["w"], gamma=rec["gamma"], beta=rec["beta"], mov_var=rec["mov_var"],
                     mU=opt["user_embedding/m"], vU=opt["user_embedding/v"], mA=opt["anime_embedding/m"],
                     vA=opt["anime_embedding/v"])
        eng.close()
    finally:
        dist.destroy_process_group()


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _check_against_oracle(tmp_path):
    d = np.load(tmp_path / "dist.npz")
    U, A, ui, ai, t, perm = _problem()
    st = orc.new_state(U, A, orc.new_head(w=1.2))
    lr, Bg = 3e-5, 2000
    losses, ns = [], []
    for k in range(0, len(perm), Bg):
        g = perm[k:k + Bg]
        met, _, _ = orc.train_step(st, ui[g], ai[g], t[g], lr)
        losses.append(float(met["loss"]) * len(g))
        ns.append(len(g))
    tol = lr * 2e-3 * len(ns)
    np.testing.assert_allclose(d["U"], st["U"], atol=tol)
    np.testing.assert_allclose(d["A"], st["A"], atol=tol)
    # the gathered Adam slots (sharded: user rows re-interleaved; replicated_rs: every rank's row shard) are the
    # single-process ones: m ~ 0.1 g per step, v ~ 1e-3 g^2
    for k in ("mU", "mA"):
        np.testing.assert_allclose(d[k], st[k], atol=2e-3 * max(np.abs(st[k]).max(), 1e-12), err_msg=k)
    for k in ("vU", "vA"):
        np.testing.assert_allclose(d[k], st[k], atol=4e-3 * max(np.abs(st[k]).max(), 1e-20), err_msg=k)
    h = st["head"]
    for k in ("w", "gamma", "beta"):
        assert abs(float(d[k]) - float(h[k])) < tol, k
    assert abs(float(d["mov_var"]) - float(h["mov_var"])) < 1e-6
    assert abs(float(d["loss"]) - sum(losses) / sum(ns)) < 5e-6
    ev = orc.evaluate(st, ui[:500], ai[:500], t[:500])
    assert abs(float(d["vl"]) - float(ev["val_loss"])) < 5e-6 and abs(float(d["vm"]) - float(ev["val_mse"])) < 1e-6


# "sharded-lazy": the user-sharded step with lazily updated user rows (sparse step + catch-up beside the all-reduce,
# a flush every 8 steps and at the end of the run) — what bench.py --gpus N runs at the S109M shape
_MODES = ["sharded", "sharded-lazy", "replicated", "replicated_rs"]


def _mode_args(mode):
    return ("sharded", True) if mode == "sharded-lazy" else (mode, False)


@pytest.mark.parametrize("mode", _MODES)
def test_two_ranks_on_one_gpu_match_oracle_on_global_batches(tmp_path, mode):
    m, lazy = _mode_args(mode)
    mp.spawn(_worker, args=(2, _port(), str(tmp_path), m, "gloo", lazy), nprocs=2, join=True)
    _check_against_oracle(tmp_path)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: two GPUs")
@pytest.mark.parametrize("mode", _MODES)
def test_two_ranks_on_two_gpus_over_rccl_match_oracle(tmp_path, mode):
    m, lazy = _mode_args(mode)
    mp.spawn(_worker, args=(2, _port(), str(tmp_path), m, "nccl", lazy), nprocs=2, join=True)
    _check_against_oracle(tmp_path)


def _nccl_world1_worker(rank, port, out_dir):
    """The N>1 step loop under the backend it ships with: RCCL, one rank, ANIREC_DIST_LOOP=1 — the collectives really
    execute (in place, on the engine's stream), from C and from torch.distributed."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                      ANIREC_DIST_LOOP="1")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from anime_recommendations_amd import schedule
        from anime_recommendations_amd.dist import DistTrainEngine
        from anime_recommendations_amd.engine import TrainEngine
        U, A, ui, ai, t, perm = _problem()
        tu, ta, tt, tp = (torch.from_numpy(np.asarray(x)).to(dev) for x in (ui, ai, t, perm))
        B = 2000
        n_steps = (len(perm) + B - 1) // B
        alphas = schedule.adam_alphas(3e-5, 1, n_steps)
        # the single-engine run (hot + rest launches, no collectives) on the same batches
        ref = TrainEngine(U.shape[0], A.shape[0], max_batch=B, arena_steps=4, device=dev)
        ref.set_head(w=1.2)
        ref.set_weights(U, A)
        starts = np.arange(n_steps) * B
        counts = np.minimum(B, len(perm) - starts)
        ref.set_epoch(tu[tp], ta[tp], tt[tp], starts, counts, alphas)
        ref.run(n_steps, use_graph=False)
        rref = ref.read_state()
        out = {}
        # both forms of the loop: inside the library with RCCL called from C (the shipping path under backend nccl),
        # and in Python with torch.distributed collectives (ANIREC_DIST_NATIVE=0)
        for mode, native in [(m, nv) for m in _MODES for nv in ("1", "graph", "0")]:
            os.environ["ANIREC_DIST_NATIVE"] = "0" if native == "0" else "1"
            os.environ["ANIREC_DIST_GRAPH"] = "1" if native == "graph" else "0"
            mode, lazy = _mode_args(mode)
            # (arena of 8: the graph variant replays two captured blocks of 4 steps, then runs the last three eagerly)
            eng = DistTrainEngine(U.shape[0], A.shape[0], B, l2=1e-4, arena_steps=8, device=dev, mode=mode, lazy=lazy)
            assert eng.loop and eng.eng.dense_mode == (1 if mode == "sharded" else 2) and eng.eng.lazy == lazy
            assert eng.native == (native != "0"), (mode, native)
            eng.set_head(w=1.2)
            eng.set_weights(U, A)
            eng.set_epoch_global(tu, ta, tt, tp, alphas)
            eng.reset_metrics()
            eng.run(n_steps)
            rec = eng.read_state()
            assert int(rec["step_fwd"]) == n_steps
            # every row sees the same fp32 operations in the same order: bit-identical tables and Adam state
            assert torch.equal(eng.eng.W, ref.W), mode
            assert torch.equal(eng.eng.M, ref.M) and torch.equal(eng.eng.V, ref.V), mode
            for k in ("w", "b", "gamma", "beta", "mov_mean", "mov_var"):
                assert rec[k] == rref[k], (mode, k)
            assert abs(float(rec["last_loss"]) - float(rref["last_loss"])) < 2e-6
            out[mode + str(lazy) + native] = eng.epoch_metrics()[0]
            eng.close()
        assert max(out.values()) - min(out.values()) < 2e-6
        ref.close()
    finally:
        dist.destroy_process_group()


def test_nccl_world1_dist_loop_equals_single_engine_bitwise(tmp_path):
    mp.spawn(_nccl_world1_worker, args=(_port(), str(tmp_path)), nprocs=1, join=True)
