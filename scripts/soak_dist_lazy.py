"""Soak of the user-sharded multi-GPU step with lazily updated user rows against the same step with dense user rows
(bitwise: the rank's user rows, the replicated anime table, the gathered Adam slots), two gloo ranks sharing cuda:0,
random problems, arena sizes and run() chunkings: python scripts/soak_dist_lazy.py [n_configs]"""
import os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, n_cfg):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from anime_recommendations_amd import schedule
    from anime_recommendations_amd.dist import DistTrainEngine
    dev = torch.device("cuda:0")
    try:
        for c in range(n_cfg):
            rng = np.random.default_rng(9000 + c)              # the same stream on both ranks
            n_u = int(rng.integers(2000, 30000)); n_a = int(rng.integers(200, 3000))
            Bl = int(rng.choice([100, 256, 500, 1000]))         # per rank
            steps = int(rng.integers(9, 50))
            n = Bl * world * steps - int(rng.integers(0, Bl))
            U = rng.uniform(-0.05, 0.05, (n_u, 128)).astype(np.float32)
            A = rng.uniform(-0.05, 0.05, (n_a, 128)).astype(np.float32)
            ui = rng.integers(0, n_u, n); ai = (rng.zipf(1.15, n) - 1) % n_a
            t = (rng.integers(0, 11, n) / 10).astype(np.float32)
            perm = rng.permutation(n)
            arena = int(rng.choice([4, 8, 16, 32]))
            chunks, left = [], steps
            while left > 0:
                k = int(min(left, rng.choice([1, 2, 3, 7, 8, 9, 17, 30])))
                chunks.append(k); left -= k
            res = {}
            for lazy in (False, True):
                eng = DistTrainEngine(n_u, n_a, Bl, l2=1e-4, arena_steps=arena, device=dev, mode="sharded", lazy=lazy)
                assert eng.eng.lazy == lazy
                eng.set_head(w=1.2); eng.set_weights(U, A)
                tu, ta, tt, tp = (torch.from_numpy(np.asarray(x)).to(dev) for x in (ui, ai, t, perm))
                eng.set_epoch_global(tu, ta, tt, tp, schedule.adam_alphas(3e-5, 1, steps))
                eng.reset_metrics()
                for k in chunks:
                    eng.run(k)
                loss, mse = eng.epoch_metrics()
                opt = eng.optimizer_state(iterations=steps)
                res[lazy] = (eng.U.cpu().numpy().copy(), eng.A.cpu().numpy().copy(), opt["user_embedding/m"].copy(),
                             opt["user_embedding/v"].copy(), float(loss), float(mse))
                eng.close()
            d, z = res[False], res[True]
            ok = all(np.array_equal(d[i], z[i]) for i in range(4)) and abs(d[4] - z[4]) <= 3e-6 * abs(d[4]) and d[5] == z[5]
            flag = torch.tensor([0 if ok else 1])
            dist.all_reduce(flag)
            if int(flag):
                if rank == 0:
                    print("MISMATCH config", c, dict(n_u=n_u, n_a=n_a, Bl=Bl, steps=steps, arena=arena, chunks=chunks), flush=True)
                sys.exit(1)
            if rank == 0 and c % 5 == 4:
                print("config %d ok" % c, flush=True)
        if rank == 0:
            print("soak ok: %d configs" % n_cfg)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port, n_cfg), nprocs=2, join=True)
