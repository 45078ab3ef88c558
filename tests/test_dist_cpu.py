"""world_size-2 gloo tests (CPU) of the multi-GPU path in its three modes (user-sharded; replicated tables with
a dense all-reduce; replicated tables with reduce-scatter -> shard Adam -> all-gather): the partition of every
global batch, and the whole DistTrainEngine step protocol (step_front -> all_gather(packets) -> step_mid ->
collective on the dense gradient -> step_back [-> all_gather(W)]) with the HIP step halves replaced by a NumPy
stand-in that speaks the same packet / dense-gradient protocol.  The distributed result must equal the single-process oracle
stepping on the GLOBAL batches."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from anime_recommendations_amd import _lib, schedule
from anime_recommendations_amd.dist import (DistTrainEngine, local_user_rows, partition_epoch,
                                             partition_epoch_replicated)
from oracle import anirec_oracle as orc

f32 = np.float32


class NumpyStageEngine:
    """Stage-level stand-in for engine.TrainEngine (tests only): same attributes, same packet and dense-gradient
    layouts and the same three step halves as libanirec, arithmetic from the oracle's formulas."""

    def __init__(self, n_user_rows, n_anime_rows, max_batch, l2=1e-4, arena_steps=64, device="cpu",
                 n_seg=1, my_seg=0, dense_mode=1, row_pad=1, adam_rows=None, lazy=None):
        self.device = torch.device("cpu")
        self.n_u, self.n_a, self.cap = n_user_rows, n_anime_rows, max_batch
        self.max_batch = max_batch
        self.rows = n_user_rows + n_anime_rows
        self.pcap = (max_batch + 3) & ~3
        self.packet_floats = 2 * self.pcap + 4
        self.n_seg, self.my_seg, self.l2 = n_seg, my_seg, f32(l2)
        self.dense_mode = dense_mode
        self.arena_steps = arena_steps
        self.packets = torch.zeros(n_seg * self.packet_floats, dtype=torch.float32)
        self.rows_alloc = (self.rows + row_pad - 1) // row_pad * row_pad
        self._Wfull = torch.zeros(self.rows_alloc, 128, dtype=torch.float32)
        self.Wn = self._Wfull.numpy()                         # shares memory with the tensor
        carried = n_anime_rows if dense_mode == 1 else self.rows
        self.dense_lo = n_user_rows if dense_mode == 1 else 0
        self.dense_rows = (carried + row_pad - 1) // row_pad * row_pad
        self.dense_grad = torch.zeros(self.dense_rows * 129, dtype=torch.float32)
        self.adam_rows = (0, 0) if adam_rows is None else tuple(adam_rows)
        self.stream = None
        self.n_steps = 0
        self.step = 0
        self.acc = dict(bce_wsum=0.0, reg_user_wsum=0.0, reg_anime_wsum=0.0, se_sum=0.0, n_seen=0.0)

    @property
    def Uw(self):
        return self.Wn[: self.n_u]

    @property
    def Aw(self):
        return self.Wn[self.n_u: self.rows]

    def set_head(self, w=1.0, **kw):
        self.hd = orc.new_head(w=w)

    def set_weights(self, U, A):
        self.Wn[: self.n_u] = np.array(U, f32)
        self.Wn[self.n_u: self.rows] = np.array(A, f32)
        self.m = np.zeros((self.rows, 128), f32)
        self.v = np.zeros((self.rows, 128), f32)

    def reset_optimizer(self):
        pass

    def set_epoch(self, lu, la, lt, starts, counts, alphas, gcounts=None):
        self.ep = (np.asarray(lu), np.asarray(la), np.asarray(lt, f32), starts, counts, np.asarray(alphas, f32))
        self.n_steps = len(counts)
        self.step = 0

    def reset_metrics(self):
        for k in self.acc:
            self.acc[k] = 0.0

    def prep(self, first, n):
        self._prepared = (first, n)

    # the stepper protocol of libanirec (anirec_dist_stepper_begin / _block): the step halves of a lazy descriptor
    # refuse to run outside a declared run and prepared block — the stand-in holds the Python loop to the same rules
    def stepper_begin(self, first_step, n_steps):
        assert first_step == self.step and 0 <= n_steps <= self.n_steps - first_step
        self._run_left, self._blk_len, self._blk_pos = n_steps, 0, 0

    def stepper_block(self, n_steps):
        assert self._prepared == (self.step, n_steps) and 0 < n_steps <= min(self.arena_steps, self._run_left)
        self._blk_len, self._blk_pos = n_steps, 0

    def _in_block(self):
        assert self._run_left > 0 and self._blk_pos < self._blk_len, "step outside a declared run / prepared block"

    def synchronize(self):
        pass

    def _batch(self):
        lu, la, lt, starts, counts, alphas = self.ep
        s, c = int(starts[self.step]), int(counts[self.step])
        return lu[s:s + c], la[s:s + c], lt[s:s + c], alphas[self.step]

    def step_front(self):
        self._in_block()
        lu, la, lt, _ = self._batch()
        u, a = self.Uw[lu], self.Aw[la]
        self.su, self.sa = np.sum(u * u, 1, dtype=f32), np.sum(a * a, 1, dtype=f32)
        ru, ra = orc._inv_norm(self.su, f32), orc._inv_norm(self.sa, f32)
        self.c = np.sum((u * ru[:, None]) * (a * ra[:, None]), 1, dtype=f32)
        pk = self.packets.numpy()[self.my_seg * self.packet_floats:(self.my_seg + 1) * self.packet_floats]
        n = len(lu)
        pk[:n] = self.c
        pk[self.pcap:self.pcap + n] = lt
        pk[2 * self.pcap:2 * self.pcap + 1].view(np.int32)[0] = n
        z = self.c * self.hd["w"] + self.hd["b"]                     # k_seg_stats: (mean, M2) of this rank's z
        pk[2 * self.pcap + 1] = np.mean(z, dtype=f32) if n else 0
        pk[2 * self.pcap + 2] = np.sum((z - pk[2 * self.pcap + 1]) ** 2, dtype=f32)

    def _head(self):
        cs, ts = [], []
        P = self.packets.numpy()
        for s in range(self.n_seg):
            pk = P[s * self.packet_floats:(s + 1) * self.packet_floats]
            n = int(pk[2 * self.pcap:2 * self.pcap + 1].view(np.int32)[0])
            cs.append(pk[:n].copy())
            ts.append(pk[self.pcap:self.pcap + n].copy())
        self.offs = np.cumsum([0] + [len(x) for x in cs])
        c, t = np.concatenate(cs), np.concatenate(ts)
        h = self.hd
        B = f32(len(c))
        z = c * h["w"] + h["b"]
        mu = np.mean(z, dtype=f32)
        var = np.mean((z - mu) ** 2, dtype=f32)
        r = f32(1) / np.sqrt(var + f32(orc.BN_EPS), dtype=f32)
        inv = r * h["gamma"]
        y = z * inv + (h["beta"] - mu * inv)
        p = orc._sigmoid(y, f32)
        dy = (p - t) / B
        zh = (z - mu) * r
        S1, S2 = np.sum(dy, dtype=f32), np.sum(dy * zh, dtype=f32)
        dzh = dy * h["gamma"]
        dz = (dzh - np.sum(dzh, dtype=f32) / B - zh * (np.sum(dzh * zh, dtype=f32) / B)) * r
        self.g_head = np.array([np.sum(dz * c, dtype=f32), np.sum(dz, dtype=f32), S2, S1], f32)
        self.glob = dict(mu=mu, var=var, n=len(c), bce=float(np.sum(orc.bce_from_logits(y, t), dtype=np.float64)),
                         se=float(np.sum((p - t) ** 2, dtype=np.float64)))
        lo, hi = self.offs[self.my_seg], self.offs[self.my_seg + 1]
        self.dc = (dz * h["w"])[lo:hi]

    def step_mid(self):
        """head + bwd + densify: sparse part of the gradient, rows >= dense_lo into the dense buffer
        ([dense_rows][128] sums of coef * other row, then [dense_rows] self-coefficient sums)."""
        self._in_block()
        self._head()
        lu, la, lt, _ = self._batch()
        ru, ra = orc._inv_norm(self.su, f32), orc._inv_norm(self.sa, f32)
        coef = self.dc * ru * ra
        self_u = np.where(self.su >= f32(orc.L2N_EPS), self.dc * self.c * ru * ru, f32(0)).astype(f32)
        self_a = np.where(self.sa >= f32(orc.L2N_EPS), self.dc * self.c * ra * ra, f32(0)).astype(f32)
        g = np.zeros((self.rows, 128), f32)
        s = np.zeros(self.rows, f32)
        np.add.at(g, lu, coef[:, None] * self.Aw[la])
        np.add.at(s, lu, self_u)
        np.add.at(g, self.n_u + la, coef[:, None] * self.Uw[lu])
        np.add.at(s, self.n_u + la, self_a)
        self.g_local, self.s_local = g, s                     # rows below dense_lo never leave the rank
        G = self.dense_grad.numpy()
        G[:] = 0
        nd, lo = self.dense_rows, self.dense_lo
        G[:nd * 128].reshape(nd, 128)[: self.rows - lo] = g[lo:]
        G[nd * 128:][: self.rows - lo] = s[lo:]

    def step_back(self):
        self._in_block()
        self._run_left -= 1
        self._blk_pos += 1
        _, _, _, alpha = self._batch()
        lo_r, hi_r = self.adam_rows if (self.adam_rows[0] | self.adam_rows[1]) else (0, self.rows)
        W = self.Wn[: self.rows]
        sq = np.sum(W.astype(np.float64) ** 2, 1)             # L2 sums of the weights this step READ,
        reg_u = float(sq[lo_r:min(hi_r, self.n_u)].sum())     # over the rows this rank's Adam covers
        reg_a = float(sq[max(lo_r, self.n_u):hi_r].sum())
        two = f32(2) * self.l2
        G = self.dense_grad.numpy()
        nd, lo = self.dense_rows, self.dense_lo
        g, s = self.g_local.copy(), self.s_local.copy()
        g[lo:] = G[:nd * 128].reshape(nd, 128)[: self.rows - lo]      # the reduced part
        s[lo:] = G[nd * 128:][: self.rows - lo]
        grad = ((g - s[:, None] * W) + two * W).astype(f32)
        sl = slice(lo_r, hi_r)
        orc.adam_update(W[sl], self.m[sl], self.v[sl], grad[sl], alpha)
        h = self.hd
        hp = np.array([h["w"], h["b"], h["gamma"], h["beta"]], f32)
        orc.adam_update(hp, h["m"], h["v"], self.g_head, alpha)
        h["w"], h["b"], h["gamma"], h["beta"] = hp
        h["mov_mean"] = f32(h["mov_mean"] - (h["mov_mean"] - self.glob["mu"]) * f32(0.01))
        h["mov_var"] = f32(h["mov_var"] - (h["mov_var"] - self.glob["var"]) * f32(0.01))
        n = self.glob["n"]
        self.acc["bce_wsum"] += self.glob["bce"]
        self.acc["reg_user_wsum"] += reg_u * n
        self.acc["reg_anime_wsum"] += reg_a * n
        self.acc["se_sum"] += self.glob["se"]
        self.acc["n_seen"] += n
        self.step += 1

    def read_state(self):
        d = dict(self.acc)
        d.update({k: self.hd[k] for k in ("w", "b", "gamma", "beta", "mov_mean", "mov_var")})
        return d

    def eval_sums(self, lu, la, lt):
        lu, la, lt = np.asarray(lu), np.asarray(la), np.asarray(lt, f32)
        lo_r, hi_r = self.adam_rows if (self.adam_rows[0] | self.adam_rows[1]) else (0, self.rows)
        sq = np.sum(self.Wn[: self.rows].astype(np.float64) ** 2, 1)
        out = dict(val_bce_sum=0.0, val_se_sum=0.0, val_n=float(len(lu)),
                   reg_user_sumsq=float(sq[lo_r:min(hi_r, self.n_u)].sum()),
                   reg_anime_sumsq=float(sq[max(lo_r, self.n_u):hi_r].sum()))
        if len(lu):
            f = orc.forward(self.Uw, self.Aw, lu, la, self.hd, training=False)
            out["val_bce_sum"] = float(np.sum(orc.bce_from_logits(f["y"], lt), dtype=np.float64))
            out["val_se_sum"] = float(np.sum((f["p"] - lt) ** 2, dtype=np.float64))
        return out

    @property
    def U(self):
        return self._Wfull[: self.n_u]

    @property
    def A(self):
        return self._Wfull[self.n_u: self.rows]

    def close(self):
        pass


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    rng = np.random.default_rng(11)
    n_u, n_a, n = 61, 40, 3 * 128 - 37
    U = rng.uniform(-0.05, 0.05, (n_u, 128)).astype(f32)
    A = rng.uniform(-0.05, 0.05, (n_a, 128)).astype(f32)
    ui = rng.integers(0, n_u, n)
    ai = (rng.zipf(1.2, n) - 1) % n_a
    t = (rng.integers(0, 11, n) / 10).astype(f32)
    perm = rng.permutation(n)
    return U, A, ui, ai, t, perm


def _worker(rank, world, port, out_dir, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        U, A, ui, ai, t, perm = _problem()
        B = 64                                   # per rank -> global batch 128
        lr = 3e-5
        tu, ta, tt, tp = (torch.from_numpy(np.asarray(x)) for x in (ui, ai, t, perm))
        # 1) the partition: my share of each global batch, in order
        if mode == "sharded":                    # by owner of the user, local rows = u // world
            lu, la, lt, starts, counts, gcounts = partition_epoch(tu, ta, tt, tp, B * world, rank, world)
            for k, (s, c) in enumerate(zip(starts, counts)):
                g = perm[k * B * world:(k + 1) * B * world]
                mine = g[ui[g] % world == rank]
                assert c == len(mine)
                assert (lu[s:s + c].numpy() == ui[mine] // world).all() and (la[s:s + c].numpy() == ai[mine]).all()
        else:                                    # contiguous slices of the batch, global user rows
            lu, la, lt, starts, counts, gcounts = partition_epoch_replicated(tu, ta, tt, tp, B * world, rank, world)
            seen = 0
            for k, (s, c) in enumerate(zip(starts, counts)):
                g = perm[k * B * world:(k + 1) * B * world]
                lo, hi = -(-len(g) * rank // world), -(-len(g) * (rank + 1) // world)
                assert c == hi - lo and c <= B
                assert (lu[s:s + c].numpy() == ui[g[lo:hi]]).all() and (la[s:s + c].numpy() == ai[g[lo:hi]]).all()
                seen += c
            assert seen == len(lu)
        assert list(gcounts) == [128, 128, len(perm) - 256]
        # 2) the step protocol
        eng = DistTrainEngine(U.shape[0], A.shape[0], B, l2=1e-4, device="cpu", engine_factory=NumpyStageEngine,
                              mode=mode)
        assert eng.n_local == (local_user_rows(U.shape[0], rank, world) if mode == "sharded" else U.shape[0])
        eng.set_head(w=1.2)
        eng.set_weights(U, A)
        n_steps = len(counts)
        eng.set_epoch_global(tu, ta, tt, tp, schedule.adam_alphas(lr, 1, n_steps))
        eng.reset_metrics()
        eng.run(n_steps)
        loss, mse = eng.epoch_metrics()
        vl, vm = eng.evaluate(tu[:100], ta[:100], tt[:100])
        Ufull = eng.U.numpy()
        if rank == 0:
            np.savez(os.path.join(out_dir, "dist.npz"), U=Ufull, A=eng.A.numpy(), loss=loss, mse=mse, vl=vl, vm=vm,
                     w=eng.read_state()["w"], gamma=eng.read_state()["gamma"])
        # replicas stay bit-identical: the anime table in every mode, the user table too when it is replicated
        for tbl in ([eng.A] if mode == "sharded" else [eng.A, eng.eng.U]):
            a_all = [torch.empty_like(tbl) for _ in range(world)]
            dist.all_gather(a_all, tbl.contiguous())
            assert all((x == a_all[0]).all() for x in a_all)
        # a share that does not fit the per-step buffers stops EVERY rank together (no rank left in a collective)
        small = DistTrainEngine(U.shape[0], A.shape[0], 4, l2=1e-4, device="cpu", engine_factory=NumpyStageEngine,
                                mode=mode)
        small.global_batch = B * world            # shares of ~64 ratings against max_batch ~ 4 + slack
        with pytest.raises(ValueError):
            small.set_epoch_global(tu, ta, tt, tp, schedule.adam_alphas(lr, 1, n_steps))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["sharded", "replicated", "replicated_rs"])
def test_two_rank_protocol_equals_single_process_oracle(tmp_path, mode):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), mode), nprocs=world, join=True)
    d = np.load(tmp_path / "dist.npz")
    U, A, ui, ai, t, perm = _problem()
    st = orc.new_state(U, A, orc.new_head(w=1.2))
    lr, Bg = 3e-5, 128
    losses, ns = [], []
    for k in range(0, len(perm), Bg):
        g = perm[k:k + Bg]
        met, _, _ = orc.train_step(st, ui[g], ai[g], t[g], lr)
        losses.append(float(met["loss"]) * len(g))
        ns.append(len(g))
    tol = lr * 2e-3 * len(ns)
    np.testing.assert_allclose(d["U"], st["U"], atol=tol)
    np.testing.assert_allclose(d["A"], st["A"], atol=tol)
    assert abs(float(d["w"]) - float(st["head"]["w"])) < tol and abs(float(d["gamma"]) - float(st["head"]["gamma"])) < tol
    assert abs(float(d["loss"]) - sum(losses) / sum(ns)) < 5e-6
    ev = orc.evaluate(st, ui[:100], ai[:100], t[:100])
    assert abs(float(d["vl"]) - float(ev["val_loss"])) < 5e-6 and abs(float(d["vm"]) - float(ev["val_mse"])) < 1e-6


def _infer_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from anime_recommendations_amd import dist_infer
        rng = np.random.default_rng(5)
        W = rng.normal(0, 0.05, (203, 128)).astype(f32)
        Wh = torch.from_numpy(orc.rownorm(W))

        def cpu_topk(What, q, k, exclude_self=True, keep=None):          # oracle stand-in for the HIP op
            i, s = orc.cosine_topk(What.numpy(), q.numpy(), k, exclude_self=exclude_self)
            return torch.from_numpy(i.astype(np.int32)), torch.from_numpy(s)

        idx, sc = dist_infer.sharded_cosine_topk(Wh, 7, topk_fn=cpu_topk)
        assert idx.shape == (203, 7)
        lo, hi = dist_infer.shard_bounds(203, rank, world)
        assert (hi - lo) in (67, 68)
        if rank == 0:
            np.savez(os.path.join(out_dir, "infer.npz"), idx=idx.numpy(), sc=sc.numpy())
    finally:
        dist.destroy_process_group()


def test_query_sharded_inference_gathers_the_single_process_result(tmp_path):
    mp.spawn(_infer_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    d = np.load(tmp_path / "infer.npz")
    W = np.random.default_rng(5).normal(0, 0.05, (203, 128)).astype(f32)
    oi, os_ = orc.cosine_topk(orc.rownorm(W), np.arange(203), 7)
    assert (d["idx"] == oi).all() and (d["sc"] == os_).all()


def _predict_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from anime_recommendations_amd import dist_infer
        U, A, head, users, watched = _predict_problem()

        def cpu_predict_topk(Ut, At, hd, us, k, wb=None):          # oracle stand-in for the HIP op
            us = np.asarray(us)
            if len(us) == 0:                                          # an empty shard: the op's early return
                return torch.empty(0, k, dtype=torch.int32), torch.empty(0, k), 0
            G = orc.predict_grid(Ut.numpy(), At.numpy(), hd, us)
            idx = np.full((len(us), k), -1, np.int32)
            p = np.full((len(us), k), np.nan, f32)
            for j in range(len(us)):
                bits = np.asarray(wb[j]).view(np.uint32)
                seen = np.array([(bits[a >> 5] >> (a & 31)) & 1 for a in range(At.shape[0])], bool)
                i, s = orc.topk_desc(G[j], k, mask=~seen)
                idx[j, :len(i)], p[j, :len(i)] = i, s
            return torch.from_numpy(idx), torch.from_numpy(p), 0

        for n_q in (7, 2):                                            # 2 users on 3 ranks: rank 2's shard is empty
            idx, p = dist_infer.sharded_predict_topk(torch.from_numpy(U), torch.from_numpy(A), head, users[:n_q], 5,
                                                     watched[:n_q], predict_fn=cpu_predict_topk)
            assert idx.shape == (n_q, 5) and p.shape == (n_q, 5)
            if rank == 0:
                np.savez(os.path.join(out_dir, "predict%d.npz" % n_q), idx=idx.numpy(), p=p.numpy())
        lo, hi = dist_infer.shard_bounds(2, 2, 3)
        assert lo == hi                                               # the empty shard was really exercised
        # the cosine path with an empty shard as well
        Wh = torch.from_numpy(orc.rownorm(A[:2]))

        def cpu_topk(What, q, k, exclude_self=True, keep=None):
            if q.numel() == 0:
                return torch.empty(0, k, dtype=torch.int32), torch.empty(0, k), 0
            i, s = orc.cosine_topk(What.numpy(), q.numpy(), k, exclude_self=exclude_self)
            return torch.from_numpy(i.astype(np.int32)), torch.from_numpy(s), 0

        i2, s2 = dist_infer.sharded_cosine_topk(Wh, 1, topk_fn=cpu_topk)
        assert i2.shape == (2, 1) and i2[:, 0].tolist() == [1, 0]
    finally:
        dist.destroy_process_group()


def _predict_problem():
    rng = np.random.default_rng(9)
    U = rng.normal(0, 0.05, (40, 128)).astype(f32)
    A = rng.normal(0, 0.05, (70, 128)).astype(f32)
    head = dict(orc.new_head(w=1.3, b=0.1, gamma=0.9, beta=-0.2, mov_mean=0.05, mov_var=0.4))
    users = np.array([3, 39, 0, 17, 17, 8, 21], np.int32)
    watched = rng.integers(0, 2 ** 32, (7, 3), dtype=np.uint64).astype(np.uint32).view(np.int32)
    return U, A, head, users, watched


def test_user_sharded_predict_topk_gathers_the_single_process_result(tmp_path):
    """dist_infer.sharded_predict_topk (model_recs.py:394-396 sharded on users) on 3 ranks,
    including a rank whose shard is empty."""
    mp.spawn(_predict_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    U, A, head, users, watched = _predict_problem()
    G = orc.predict_grid(U, A, head, users)
    for n_q in (7, 2):
        d = np.load(tmp_path / ("predict%d.npz" % n_q))
        for j in range(n_q):
            bits = watched[j].view(np.uint32)
            seen = np.array([(bits[a >> 5] >> (a & 31)) & 1 for a in range(A.shape[0])], bool)
            i, s = orc.topk_desc(G[j], 5, mask=~seen)
            assert d["idx"][j].tolist() == list(i) and np.array_equal(d["p"][j], s.astype(f32))
