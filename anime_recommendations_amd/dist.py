"""Multi-GPU training: one process per MI355X, ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" for CPU/1-GPU rehearsal).

Three partitionings, selected by ``ANIREC_DP_MODE`` (or the ``mode`` argument):

``sharded`` (default; DESIGN.md "Multi-GPU") — ratings are sharded data-parallel BY USER: rank r owns
the users u with u % G == r, their embedding rows and Adam moments (local row u // G), and processes
exactly the ratings of those users in every global batch.  Consequences:
  * the user-table gradient of a rating always lands on the rank that computed it: 95 % of
    the parameters (350 k of 368 k rows at the 109 M shape) need NO gradient exchange and the
    dense Adam stream over them shrinks by G;
  * the anime table (18 k rows, 9.2 MB) is replicated; its dense gradient is summed with ONE
    RCCL all-reduce per step — the "dense embedding-gradient all-reduce" of the north star,
    applied to the only table whose gradient is shared.

``replicated`` — the reference's own data-parallel construct taken literally
(neural_network/neural_network.py:173-178: replicated variables under TPUStrategy): both tables and the
Adam state replicated on every rank, each global batch cut into G contiguous slices, the dense gradient
of BOTH tables ((n_users + n_anime) x 128 floats: 188 MB at the 109 M shape) all-reduced every step,
every rank running the whole dense Adam.  This is the baseline SURVEY §7-H3 names; it cannot scale
(the all-reduce alone costs more than the 1-GPU step) and exists to be reported next to the others.

``replicated_rs`` — the same replicated tables with the collective cut in two: reduce-scatter of the dense
gradient -> each rank runs Adam on its 1/G row shard -> all-gather of the updated rows (same wire bytes as
the all-reduce, Adam traffic divided by G).

In every mode BatchNorm couples the global batch: one all-gather of the 8-byte-per-rating head packets
(c, t) per step lets every rank redo the cheap scalar head over the whole batch, so the result equals a
single-GPU step on the global batch (not per-replica BN).

The step:
    step_front (fwd)  ->  all_gather(packets)  ->  step_mid (head, bwd, densify [+ user-row adam forked onto
    a side stream])  ->  all_reduce / reduce_scatter(dense grad)  ->  step_back (adam of the rows that needed
    the collective + step finish)  [->  all_gather(W) in replicated_rs].
Backend "nccl": the whole loop runs INSIDE libanirec (``anirec_dist_run``: one C call per run(), the collectives issued
to RCCL from C on the engine's stream through the library's own communicator, bootstrapped with a unique id that
torch.distributed broadcasts once).  Any other backend (gloo: CPU / one-GPU rehearsal) keeps the loop in Python with
torch.distributed collectives — three C calls and two or three collectives per step.  The path is chosen ONCE, at
construction, from ``dist.get_backend()`` (RCCL reports errors asynchronously: a try/except around a collective cannot
be the thing that saves a step).
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib

MODES = ("sharded", "replicated", "replicated_rs")


def local_user_rows(n_users, rank, world):
    return (n_users - rank + world - 1) // world


def batch_slack(batch):
    """Per-rank share of a global batch is Binomial(G*B, 1/G): mean B, sd < sqrt(B)."""
    return int(batch + 6 * np.sqrt(batch) + 16)


def _step_tables(n, global_batch, pos, device):
    n_steps = (n + global_batch - 1) // global_batch
    counts = torch.bincount(pos // global_batch, minlength=n_steps).cpu().numpy().astype(np.int64)
    starts = np.cumsum(counts) - counts
    gstarts = np.arange(n_steps) * global_batch
    gcounts = np.minimum(global_batch, n - gstarts)
    return starts, counts, gcounts


def partition_epoch(user, anime, rating, perm, global_batch, rank, world):
    """This rank's ratings of every global batch, in global (shuffled) order — user-sharded mode.

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
    starts, counts, gcounts = _step_tables(n, global_batch, pos, perm.device)
    lu = (pu[pos] // world).to(torch.int32)
    return lu, anime[idx].to(torch.int32), rating[idx].to(torch.float32), starts, counts, gcounts


def partition_epoch_replicated(user, anime, rating, perm, global_batch, rank, world):
    """Replicated-table modes: batch k = perm[k*Bg:(k+1)*Bg] is cut into ``world`` contiguous slices whose
    sizes differ by at most one; this rank takes slice ``rank``.  User indices stay global rows."""
    n = perm.numel()
    p = torch.arange(n, device=perm.device)
    k = p // global_batch
    off = p - k * global_batch
    cnt = torch.clamp(n - k * global_batch, max=global_batch)
    pos = torch.nonzero((off * world) // cnt == rank, as_tuple=False).flatten()
    idx = perm[pos]
    starts, counts, gcounts = _step_tables(n, global_batch, pos, perm.device)
    return (user[idx].to(torch.int32), anime[idx].to(torch.int32), rating[idx].to(torch.float32),
            starts, counts, gcounts)


class DistTrainEngine:
    """TrainEngine facade for G ranks (same interface as engine.TrainEngine for trainer.fit)."""

    def __init__(self, n_users, n_anime, batch_per_rank, l2=1e-4, arena_steps=64, device="cuda:0",
                 engine_factory=None, mode=None, lazy=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        if self.world > _lib.MAX_SEG:
            raise ValueError("at most %d ranks" % _lib.MAX_SEG)
        self.mode = mode or os.environ.get("ANIREC_DP_MODE", "sharded")
        if self.mode not in MODES:
            raise ValueError("ANIREC_DP_MODE must be one of %s (got %r)" % (MODES, self.mode))
        self.n_users, self.n_anime = int(n_users), int(n_anime)
        self.batch_per_rank = int(batch_per_rank)
        self.global_batch = self.batch_per_rank * self.world
        if engine_factory is None:
            from .engine import TrainEngine as engine_factory
        # ANIREC_DIST_LOOP=1: rehearse the N>1 loop on one rank; the replicated modes only exist as that loop
        loop = self.world > 1 or os.environ.get("ANIREC_DIST_LOOP") == "1" or self.mode != "sharded"
        self.loop = loop
        if self.mode == "sharded":
            self.n_local = local_user_rows(self.n_users, self.rank, self.world)
            max_batch = min(_lib.MAX_BATCH, batch_slack(self.batch_per_rank))
            # lazy: the lazy dense Adam — of the whole step on one rank without the loop, of this rank's user rows in
            # the user-sharded loop (None: the engine decides from the shard size; ANIREC_LAZY_ADAM overrides)
            kw = dict(dense_mode=1 if loop else 0, lazy=lazy)
        else:
            self.n_local = self.n_users
            max_batch = self.batch_per_rank
            kw = dict(dense_mode=2)
            if self.mode == "replicated_rs":
                rows = self.n_users + self.n_anime
                self.shard_rows = (rows + self.world - 1) // self.world
                lo = min(rows, self.rank * self.shard_rows)
                kw.update(row_pad=self.world, adam_rows=(lo, min(rows, lo + self.shard_rows)))
        self.eng = engine_factory(self.n_local, self.n_anime, max_batch=max_batch, l2=l2,
                                  arena_steps=arena_steps, device=device, n_seg=self.world,
                                  my_seg=self.rank, **kw)
        self.device = self.eng.device
        self.l2 = float(l2)
        self.cursor = 0
        pf = self.eng.packet_floats
        self._send = torch.zeros(pf, dtype=torch.float32, device=self.device)
        # collective variants of the PYTHON loop, decided once from the backend: RCCL has the flat-tensor forms and
        # reduce-scatter; gloo has neither (list all-gather, and the shard sums by all-reduce)
        backend = dist.get_backend()
        self._use_flat_gather = backend == "nccl"
        self._use_reduce_scatter = backend == "nccl"
        self._comm = None
        if loop and backend == "nccl" and os.environ.get("ANIREC_DIST_NATIVE", "0") == "1":
            self._comm = self._native_comm()
        if self.mode == "replicated_rs":
            sr = self.shard_rows
            self._rs_g = torch.zeros(sr * _lib.DIM, dtype=torch.float32, device=self.device)
            self._rs_s = torch.zeros(sr, dtype=torch.float32, device=self.device)
            self._ag_w = torch.zeros(sr, _lib.DIM, dtype=torch.float32, device=self.device)

    def _native_comm(self):
        """The library's own RCCL communicator over the same ranks (collective).  Every rank must take the same
        decision: the outcome of loading RCCL is agreed with one MIN all-reduce before anyone creates a communicator."""
        import ctypes as C
        import sys
        lib = self.eng.lib
        ok = int(lib.anirec_rccl_load(None) == 0)
        flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 0:
            return None
        # RCCL prints a version banner on STDOUT when it is initialised outside torch: keep stdout clean (bench.py's
        # contract is ONE JSON line there) by pointing fd 1 at stderr while the id and the communicator are made
        sys.stdout.flush()
        saved = os.dup(1)
        h = C.c_void_p()
        try:
            os.dup2(2, 1)
            idb = C.create_string_buffer(_lib.RCCL_ID_BYTES)
            rc = lib.anirec_rccl_unique_id(idb) if self.rank == 0 else 0
            # the id + rank 0's status in one broadcast: if rank 0 failed, EVERY rank raises, nobody is left in a collective
            t = torch.frombuffer(bytearray(idb.raw) + bytearray([1 if rc == 0 else 0]), dtype=torch.uint8).to(self.device)
            dist.broadcast(t, src=0)
            got = t.cpu().numpy().tobytes()
            raw = got[:_lib.RCCL_ID_BYTES]
            if rc == 0 and got[-1] != 1:
                rc = -5                                      # ANIREC_ECOMM: rank 0 could not make the id
            if rc == 0:
                with torch.cuda.device(self.device):
                    rc = lib.anirec_dist_comm_create(raw, self.rank, self.world, C.byref(h))
        finally:
            try:
                C.CDLL(None).fflush(None)     # RCCL printf()s into libc's buffer: drain it while fd 1 is still stderr
            except OSError:
                pass
            os.dup2(saved, 1)
            os.close(saved)
        _lib.check(rc, "anirec_rccl_unique_id / anirec_dist_comm_create")
        return h

    @property
    def native(self):
        """True when run() is one C call with RCCL inside (backend nccl), False for the Python step loop."""
        return self._comm is not None

    # ---- weights ---------------------------------------------------------------------------
    def set_head(self, **kw):
        self.eng.set_head(**kw)

    def set_weights(self, U, A):
        U = torch.as_tensor(U, dtype=torch.float32)
        self.eng.set_weights(U[self.rank::self.world].contiguous() if self.mode == "sharded" else U, A)

    def reset_optimizer(self):
        self.eng.reset_optimizer()

    @property
    def A(self):
        return self.eng.A

    @property
    def U(self):
        """Full user table (sharded mode: all-gathered and re-interleaved; collective call)."""
        return self.gather_user_table() if self.mode == "sharded" else self.eng.U

    def optimizer_state(self, iterations=0):
        """Collective: the full-table Adam slots."""
        st = self.eng.optimizer_state(iterations)
        nl = self.n_local
        if self.mode == "sharded":
            st["user_embedding/m"] = self.gather_user_table(self.eng.M[:nl]).cpu().numpy()
            st["user_embedding/v"] = self.gather_user_table(self.eng.V[:nl]).cpu().numpy()
        elif self.mode == "replicated_rs" and self.world > 1:
            # every rank only keeps the moments of its own row shard up to date
            for key, t in (("m", self.eng.M), ("v", self.eng.V)):
                full = self._gather_row_shards(t)
                st["user_embedding/" + key] = full[:nl].cpu().numpy()
                st["anime_embedding/" + key] = full[nl:].cpu().numpy()
        return st

    def _gather_row_shards(self, t):
        rows, sr = self.n_users + self.n_anime, self.shard_rows
        lo, hi = self.eng.adam_rows
        mine = torch.zeros(sr, _lib.DIM, dtype=torch.float32, device=self.device)
        mine[: hi - lo] = t[lo:hi]
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine)
        return torch.cat(parts, 0)[:rows]

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
        part = partition_epoch if self.mode == "sharded" else partition_epoch_replicated
        lu, la, lt, starts, counts, gcounts = part(user, anime, rating, perm, self.global_batch, self.rank, self.world)
        # a share that overflows the per-step buffers must stop EVERY rank, not leave the others in a collective
        worst = torch.tensor([int(counts.max()) if len(counts) else 0], dtype=torch.int64, device=self.device)
        if self.world > 1:
            dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        if int(worst[0]) > self.eng.max_batch:
            raise ValueError("a rank's share of a global batch holds %d ratings > max_batch %d (per-rank batch %d)"
                             % (int(worst[0]), self.eng.max_batch, self.batch_per_rank))
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
            dist.all_gather_into_tensor(e.packets, self._send)
        else:
            dist.all_gather([e.packets[r * pf:(r + 1) * pf] for r in range(self.world)], self._send)

    def _reduce_dense(self):
        """Sum the dense gradient buffer over the ranks (RCCL over xGMI): all-reduce, or — replicated_rs —
        reduce-scatter so that each rank only receives the rows its Adam shard updates."""
        e = self.eng
        g = e.dense_grad
        if self.mode != "replicated_rs":
            dist.all_reduce(g)
            return
        nd, sr = e.dense_rows, self.shard_rows
        if self._use_reduce_scatter:
            dist.reduce_scatter_tensor(self._rs_g, g[: nd * _lib.DIM])
            dist.reduce_scatter_tensor(self._rs_s, g[nd * _lib.DIM:])
            lo = self.rank * sr
            g[lo * _lib.DIM:(lo + sr) * _lib.DIM].copy_(self._rs_g)
            g[nd * _lib.DIM + lo: nd * _lib.DIM + lo + sr].copy_(self._rs_s)
        else:                                               # gloo has no reduce-scatter: the same sums by all-reduce
            dist.all_reduce(g)

    def _all_gather_rows(self):
        """replicated_rs: every rank updated its row shard; collect the updated rows of W."""
        e = self.eng
        sr = self.shard_rows
        lo = self.rank * sr
        self._ag_w.copy_(e._Wfull[lo:lo + sr])
        if self._use_flat_gather:
            dist.all_gather_into_tensor(e._Wfull, self._ag_w)
        else:
            dist.all_gather([e._Wfull[r * sr:(r + 1) * sr] for r in range(self.world)], self._ag_w)

    def step(self):
        """One optimiser step of the multi-GPU loop: 3 C calls + 2 (3) collectives."""
        e = self.eng
        e.step_front()                      # fwd (+ this rank's BatchNorm partial statistics)
        self._all_gather_packets()          # (c, t, count) of every rank: BatchNorm sees the global batch
        e.step_mid()                        # head, bwd, densify; sharded: user-row adam forked beside the collective
        self._reduce_dense()
        e.step_back()                       # adam of the rows that needed the sum + step finish
        if self.mode == "replicated_rs":
            self._all_gather_rows()

    def run(self, n_steps=None, use_graph=True, first_step=None):
        e = self.eng
        if first_step is None:
            first_step = self.cursor
        if n_steps is None:
            n_steps = e.n_steps - first_step
        if not self.loop:
            e.run(n_steps, use_graph=use_graph, first_step=first_step)
            self.cursor = first_step + n_steps
            return n_steps
        if self._comm is not None:              # the whole loop in the library, RCCL called from C
            # ANIREC_DIST_GRAPH=1: replay captured blocks of steps (RCCL inside the hipGraph; rehearsed with one rank,
            # never yet run on several GPUs: opt-in)
            graph = int(bool(use_graph) and os.environ.get("ANIREC_DIST_GRAPH", "0") == "1")
            _lib.check(e.lib.anirec_dist_run(e._get_stepper(), self._comm, int(first_step), int(n_steps), graph,
                                             e._sp()), "anirec_dist_run")
            self.cursor = first_step + n_steps
            return n_steps
        done = 0
        with torch.cuda.stream(e.stream) if self.device.type == "cuda" else _null_ctx():
            e.stepper_begin(first_step, n_steps)       # (lazy user rows: a window opens here, the last step flushes)
            while done < n_steps:
                blk = min(e.arena_steps, n_steps - done)
                e.prep(first_step + done, blk)
                e.stepper_block(blk)
                for _ in range(blk):
                    self.step()
                done += blk
        self.cursor = first_step + n_steps
        return n_steps

    # ---- metrics ---------------------------------------------------------------------------
    def _sum_over_ranks(self, values):
        t = torch.tensor(values, dtype=torch.float64, device=self.device)
        if self.world > 1:
            dist.all_reduce(t)
        return [float(x) for x in t]

    def epoch_metrics(self):
        rec = self.eng.read_state()
        ru, ra = float(rec["reg_user_wsum"]), float(rec["reg_anime_wsum"])
        if self.mode == "sharded":          # user rows are sharded, the anime table is whole on every rank
            ru = self._sum_over_ranks([ru])[0]
        elif self.mode == "replicated_rs":  # each rank's Adam (and so its L2 partials) covers one row shard
            ru, ra = self._sum_over_ranks([ru, ra])
        n = max(float(rec["n_seen"]), 1.0)
        return (float(rec["bce_wsum"]) + self.l2 * (ru + ra)) / n, float(rec["se_sum"]) / n

    def evaluate(self, user, anime, rating):
        """Validation rows are split over the ranks (sharded: by owner of the user); sums all-reduced."""
        user = torch.as_tensor(user, device=self.device).to(torch.int64)
        if self.mode == "sharded":
            mine = (user % self.world) == self.rank
            lu = (user[mine] // self.world).to(torch.int32)
        else:
            mine = (torch.arange(user.numel(), device=self.device) % self.world) == self.rank
            lu = user[mine].to(torch.int32)
        la = torch.as_tensor(anime, device=self.device)[mine].to(torch.int32)
        lt = torch.as_tensor(rating, device=self.device)[mine].to(torch.float32)
        rec = self.eng.eval_sums(lu, la, lt)             # L2 sums: of the rows this rank's Adam updates
        bce, se, n = self._sum_over_ranks([rec["val_bce_sum"], rec["val_se_sum"], rec["val_n"]])
        ru, ra = float(rec["reg_user_sumsq"]), float(rec["reg_anime_sumsq"])
        if self.mode == "sharded":
            ru = self._sum_over_ranks([ru])[0]
        elif self.mode == "replicated_rs":
            ru, ra = self._sum_over_ranks([ru, ra])
        n = max(n, 1.0)
        return bce / n + self.l2 * (ru + ra), se / n

    def read_state(self):
        return self.eng.read_state()

    def synchronize(self):
        self.eng.synchronize()

    def close(self):
        if self._comm is not None:
            self.eng.synchronize()
            self.eng.lib.anirec_dist_comm_destroy(self._comm)
            self._comm = None
        self.eng.close()


class _null_ctx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
