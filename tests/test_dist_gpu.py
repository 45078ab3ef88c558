"""GPU test of the multi-GPU path with the REAL HIP stages: two ranks share cuda:0 and talk over
gloo (RCCL needs one GPU per rank; the driver exercises that on the 8-GPU node).  The two-rank
result must match the oracle stepping on the global batches, and the replicated anime table
must stay bit-identical across ranks."""
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
    n_u, n_a, n = 3001, 700, 5 * 2000 - 333
    U = rng.uniform(-0.05, 0.05, (n_u, 128)).astype(np.float32)
    A = rng.uniform(-0.05, 0.05, (n_a, 128)).astype(np.float32)
    ui = rng.integers(0, n_u, n)
    ai = (rng.zipf(1.15, n) - 1) % n_a
    t = (rng.integers(0, 11, n) / 10).astype(np.float32)
    return U, A, ui, ai, t, rng.permutation(n)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from anime_recommendations_amd import schedule
        from anime_recommendations_amd.dist import DistTrainEngine
        U, A, ui, ai, t, perm = _problem()
        dev = torch.device("cuda:0")
        eng = DistTrainEngine(U.shape[0], A.shape[0], 1000, l2=1e-4, arena_steps=4, device=dev)
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
        a_all = [torch.empty_like(Aloc) for _ in range(world)]
        dist.all_gather(a_all, Aloc)
        assert all((x == a_all[0]).all() for x in a_all)
        if rank == 0:
            rec = eng.read_state()
            np.savez(os.path.join(out_dir, "dist.npz"), U=Ufull, A=Aloc.numpy(), loss=loss, mse=mse, vl=vl, vm=vm,
                     w=rec["w"], gamma=rec["gamma"], beta=rec["beta"], mov_var=rec["mov_var"])
        eng.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_oracle_on_global_batches(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
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
    h = st["head"]
    for k in ("w", "gamma", "beta"):
        assert abs(float(d[k]) - float(h[k])) < tol, k
    assert abs(float(d["mov_var"]) - float(h["mov_var"])) < 1e-6
    assert abs(float(d["loss"]) - sum(losses) / sum(ns)) < 5e-6
    ev = orc.evaluate(st, ui[:500], ai[:500], t[:500])
    assert abs(float(d["vl"]) - float(ev["val_loss"])) < 5e-6 and abs(float(d["vm"]) - float(ev["val_mse"])) < 1e-6
