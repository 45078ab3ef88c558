"""Multi-GPU training: one process per MI355X, ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" for CPU/1-GPU rehearsal).

Partitioning (DESIGN.md "Multi-GPU"): ratings are sharded data-parallel BY USER — rank r owns
the users u with u % G == r, their embedding rows and Adam moments (local row u // G), and
processes exactly the ratings of those users in every global batch.  Consequences:
  * the user-table gradient of a rating always lands on the rank that computed it: 95 % of
    the parameters (350 k of 368 k rows at the 109 M shape) need NO gradient exchange and the
    dense Adam stream over them shrinks by G;
  * the anime table (18 k rows, 9.2 MB) is replicated; its dense gradient is summed with ONE
    RCCL all-reduce per step — the "dense embedding-gradient all-reduce" of the north star,
    applied to the only table whose gradient is shared;
  * BatchNorm couples the global batch: one all-gather of the 8-byte-per-rating head packets
    (c, t) per step lets every rank redo the cheap scalar head over the whole batch, so the
    result equals a single-GPU step on the global batch (not per-replica BN).
Per step: fwd -> all_gather(packets) -> head -> bwd -> all_reduce(anime grad) -> adam.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


def local_user_rows(n_users, rank, world):
    return (n_users - rank + world - 1) // world


def batch_slack(batch):
    """Per-rank share of a global batch is Binomial(G*B, 1/G): mean B, sd < sqrt(B)."""
    return int(batch + 6 * np.sqrt(batch) + 16)


def partition_epoch(user, anime, rating, perm, global_batch, rank, world):
    """This rank's ratings of every global batch, in global (shuffled) order.

    user/anime/rating: the full training columns (torch, any device); perm: epoch permutation.
    Returns (local_user_row int32, anime int32, rating fp32, starts, counts, global_counts)
    where batch k of the epoch is perm[k*global_batch:(k+1)*global_batch] and this rank holds
    the entries whose user % world == rank (local row = user // world).
    """
    n = perm.numel()
    pu = user[perm].to(torch.int64)
    mine = (pu % world) == rank
    pos = torch.nonzero(mine, as_tuple=False).flatten()
    idx = perm[pos]
    n_steps = (n + global_batch - 1) // global_batch
    counts = torch.bincount(pos // global_batch, minlength=n_steps).cpu().numpy().astype(np.int64)
    starts = np.cumsum(counts) - counts
    gstarts = np.arange(n_steps) * global_batch
    gcounts = np.minimum(global_batch, n - gstarts)
    lu = (pu[pos] // world).to(torch.int32)
    return lu, anime[idx].to(torch.int32), rating[idx].to(torch.float32), starts, counts, gcounts


class DistTrainEngine:
    """TrainEngine facade for G ranks (same interface as engine.TrainEngine for trainer.fit)."""

    def __init__(self, n_users, n_anime, batch_per_rank, l2=1e-4, arena_steps=64, device="cuda:0",
                 engine_factory=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        if self.world > _lib.MAX_SEG:
            raise ValueError("at most %d ranks" % _lib.MAX_SEG)
        self.n_users, self.n_anime = int(n_users), int(n_anime)
        self.batch_per_rank = int(batch_per_rank)
        self.global_batch = self.batch_per_rank * self.world
        self.n_local = local_user_rows(self.n_users, self.rank, self.world)
        max_batch = min(_lib.MAX_BATCH, batch_slack(self.batch_per_rank))
        if engine_factory is None:
            from .engine import TrainEngine as engine_factory
        self.eng = engine_factory(self.n_local, self.n_anime, max_batch=max_batch, l2=l2,
                                  arena_steps=arena_steps, device=device, n_seg=self.world,
                                  my_seg=self.rank,
                                  anime_dense=self.world > 1 or os.environ.get("ANIREC_DIST_LOOP") == "1")
        self.device = self.eng.device
        self.l2 = float(l2)
        self.cursor = 0
        pf = self.eng.packet_floats
        self._send = torch.zeros(pf, dtype=torch.float32, device=self.device)
        self._use_flat_gather = True

    # ---- weights ---------------------------------------------------------------------------
    def set_head(self, **kw):
        self.eng.set_head(**kw)

    def set_weights(self, U, A):
        U = torch.as_tensor(U, dtype=torch.float32)
        self.eng.set_weights(U[self.rank::self.world].contiguous(), A)

    def reset_optimizer(self):
        self.eng.reset_optimizer()

    @property
    def A(self):
        return self.eng.A

    @property
    def U(self):
        """Full user table (all-gathered and re-interleaved); collective call."""
        return self.gather_user_table()

    def optimizer_state(self, iterations=0):
        """Collective: the full-table Adam slots (user rows re-interleaved from the ranks' shards)."""
        st = self.eng.optimizer_state(iterations)
        nl = self.n_local
        st["user_embedding/m"] = self.gather_user_table(self.eng.M[:nl]).cpu().numpy()
        st["user_embedding/v"] = self.gather_user_table(self.eng.V[:nl]).cpu().numpy()
        return st

    def gather_user_table(self, local=None):
        self.eng.synchronize()
        n_max = local_user_rows(self.n_users, 0, self.world)
        mine = torch.zeros(n_max, _lib.DIM, dtype=torch.float32, device=self.device)
        mine[: self.n_local] = self.eng.U if local is None else local
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine)
        full = torch.empty(self.n_users, _lib.DIM, dtype=torch.float32, device=self.device)
        for r in range(self.world):
            full[r::self.world] = parts[r][: local_user_rows(self.n_users, r, self.world)]
        return full

    # ---- epoch -----------------------------------------------------------------------------
    def set_epoch_global(self, user, anime, rating, perm, alphas):
        """Install this rank's share of an epoch given the FULL training columns and the
        (identical on every rank) epoch permutation."""
        lu, la, lt, starts, counts, gcounts = partition_epoch(user, anime, rating, perm, self.global_batch,
                                                              self.rank, self.world)
        self.eng.set_epoch(lu, la, lt, starts, counts, alphas, gcounts)
        self.cursor = 0
        return len(counts)

    def reset_metrics(self):
        self.eng.reset_metrics()

    # ---- the step loop ---------------------------------------------------------------------
    def _all_gather_packets(self):
        e = self.eng
        pf = e.packet_floats
        self._send.copy_(e.packets[self.rank * pf:(self.rank + 1) * pf])
        if self._use_flat_gather:
            try:
                dist.all_gather_into_tensor(e.packets, self._send)
                return
            except (RuntimeError, NotImplementedError):
                self._use_flat_gather = False
        dist.all_gather([e.packets[r * pf:(r + 1) * pf] for r in range(self.world)], self._send)

    def run(self, n_steps=None, use_graph=True, first_step=None):
        e = self.eng
        if first_step is None:
            first_step = self.cursor
        if n_steps is None:
            n_steps = e.n_steps - first_step
        if self.world == 1 and os.environ.get("ANIREC_DIST_LOOP") != "1":   # (=1: rehearse the N>1 loop on one rank)
            e.run(n_steps, use_graph=use_graph, first_step=first_step)
            self.cursor = first_step + n_steps
            return n_steps
        done = 0
        with torch.cuda.stream(e.stream) if self.device.type == "cuda" else _null_ctx():
            while done < n_steps:
                blk = min(e.arena_steps, n_steps - done)
                e.prep(first_step + done, blk)
                for _ in range(blk):
                    e.fwd()
                    self._all_gather_packets()          # (c, t, count) of every rank: BatchNorm sees the global batch
                    e.head()
                    e.bwd()                              # user chunks stay local; anime gradient densified
                    # RCCL sum over xGMI of the one shared table, overlapped with the dense Adam
                    # stream over this rank's user rows (which does not need it)
                    work = dist.all_reduce(e.anime_grad, async_op=True)
                    e.adam_users()
                    work.wait()
                    e.adam_anime_finish()
                done += blk
        self.cursor = first_step + n_steps
        return n_steps

    # ---- metrics ---------------------------------------------------------------------------
    def epoch_metrics(self):
        rec = self.eng.read_state()
        t = torch.tensor([float(rec["reg_user_wsum"])], dtype=torch.float64, device=self.device)
        if self.world > 1:
            dist.all_reduce(t)
        n = max(float(rec["n_seen"]), 1.0)
        loss = (float(rec["bce_wsum"]) + self.l2 * (float(t[0]) + float(rec["reg_anime_wsum"]))) / n
        return loss, float(rec["se_sum"]) / n

    def evaluate(self, user, anime, rating):
        """Validation rows are evaluated by the rank owning their user; sums all-reduced."""
        user = torch.as_tensor(user, device=self.device).to(torch.int64)
        mine = (user % self.world) == self.rank
        lu = (user[mine] // self.world).to(torch.int32)
        la = torch.as_tensor(anime, device=self.device)[mine].to(torch.int32)
        lt = torch.as_tensor(rating, device=self.device)[mine].to(torch.float32)
        rec = self.eng.eval_sums(lu, la, lt)
        t = torch.tensor([rec["val_bce_sum"], rec["val_se_sum"], rec["val_n"], rec["reg_user_sumsq"]],
                         dtype=torch.float64, device=self.device)
        if self.world > 1:
            dist.all_reduce(t)
        n = max(float(t[2]), 1.0)
        val_loss = float(t[0]) / n + self.l2 * (float(t[3]) + float(rec["reg_anime_sumsq"]))
        return val_loss, float(t[1]) / n

    def read_state(self):
        return self.eng.read_state()

    def synchronize(self):
        self.eng.synchronize()

    def close(self):
        self.eng.close()


class _null_ctx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
