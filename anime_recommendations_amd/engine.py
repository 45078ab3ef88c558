"""Host-side owner of the device memory behind the training hot path.

PyTorch is used here for plumbing only: HBM allocations, streams, (later) RCCL.
All arithmetic of the train step runs in libanirec's HIP kernels.

HBM layout (one rank):
  W, M, V     [(n_user_rows + n_anime_rows), 128] fp32 each — embeddings and Adam moments,
              users first then anime, so one dense Adam launch covers both tables
  rowmap      [2][rows] int32 — per-step "row -> chunk list" map written by bwd, cleared by adam (one per step parity)
  state       anirec_state (136 B) — scalar head, BN moving stats, step cursors, metrics
  epoch data  user_idx/anime_idx int32 + rating fp32 in shuffled epoch order (12 B/rating)
  sched       anirec_step[n_steps] — start/count/alpha of every step of the epoch
  workspace   per-step scratch + the prep arena (sorted batches + chunk tables)
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import DIM


def _dev_bytes(n, device):
    return torch.zeros(int(n), dtype=torch.uint8, device=device)


class TrainEngine:
    def __init__(self, n_user_rows, n_anime_rows, max_batch, l2=1e-4, arena_steps=64,
                 device="cuda:0", n_seg=1, my_seg=0, dense_mode=0, row_pad=1, adam_rows=None, lazy=None):
        """dense_mode: 0 one GPU; 1 user-sharded DP (anime gradient through ``dense_grad``); 2 replicated
        tables (every gradient through ``dense_grad``).  row_pad: the tables and the dense buffer are
        allocated with their row count rounded up to a multiple of it (equal reduce-scatter / all-gather
        shards).  adam_rows: (lo, hi) row shard this rank's Adam updates in mode 2, None = all.
        lazy: the lazy dense Adam (include/anirec.h: rows a batch does not touch take their L2-only steps later,
        several at a time; tables and Adam state bit-identical to the dense update) — of ``run`` on one GPU (both
        tables) and of the user-sharded multi-GPU step (dense_mode 1: this rank's user rows; the replicated anime
        rows keep their dense update behind the all-reduce).  None = automatic: tables of at least 8 batches' worth of
        rows on one GPU, user shards of at least 6 in mode 1 (below that most rows are touched every few steps and
        the plain dense stream is faster); ANIREC_LAZY_ADAM=0/1 overrides.  Never in dense_mode 2: that step is
        bound by its 188 MB collectives, not by the Adam stream."""
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.AnirecError("no GPU: the anime_recommendations_amd hot path needs an MI355X")
        if not (1 <= max_batch <= _lib.MAX_BATCH):
            raise ValueError("max_batch must be in 1..%d (got %d)" % (_lib.MAX_BATCH, max_batch))
        self.device = torch.device(device)
        self.n_user_rows, self.n_anime_rows = int(n_user_rows), int(n_anime_rows)
        self.rows = self.n_user_rows + self.n_anime_rows
        self.max_batch, self.arena_steps = int(max_batch), int(arena_steps)
        self.l2 = float(l2)
        self.n_seg, self.my_seg, self.dense_mode = int(n_seg), int(my_seg), int(dense_mode)
        dev = self.device
        row_pad = max(1, int(row_pad))
        self.rows_alloc = (self.rows + row_pad - 1) // row_pad * row_pad
        self._Wfull = torch.zeros(self.rows_alloc, DIM, dtype=torch.float32, device=dev)
        self._W = self._Wfull[: self.rows]
        self._M = torch.zeros(self.rows, DIM, dtype=torch.float32, device=dev)
        self._V = torch.zeros_like(self._M)
        self.rowmap = torch.zeros(2 * self.rows, dtype=torch.int32, device=dev)
        self.adam_rows = (0, 0) if adam_rows is None else (int(adam_rows[0]), int(adam_rows[1]))
        self.state_buf = _dev_bytes(_lib.STATE_DTYPE.itemsize, dev)
        self.packet_floats = int(self.lib.anirec_packet_floats(self.max_batch))
        self.packets = torch.zeros(self.n_seg * self.packet_floats, dtype=torch.float32, device=dev)
        ws = int(self.lib.anirec_train_workspace_bytes(self.max_batch, self.arena_steps))
        if ws == 0:
            raise _lib.AnirecError("anirec_train_workspace_bytes rejected the geometry")
        self.workspace = _dev_bytes(ws, dev)
        # dense gradient buffer of the multi-GPU modes: [dense_rows][128] gradients, then [dense_rows] self sums
        carried = {0: 0, 1: self.n_anime_rows, 2: self.rows}[self.dense_mode]
        self.dense_rows = (carried + row_pad - 1) // row_pad * row_pad
        self.dense_grad = (torch.zeros(self.dense_rows * (DIM + 1), dtype=torch.float32, device=dev)
                           if self.dense_mode else None)
        if lazy is None:
            import os
            env = os.environ.get("ANIREC_LAZY_ADAM")
            big = (self.rows >= 8 * self.max_batch) if self.dense_mode == 0 else (self.n_user_rows >= 6 * self.max_batch)
            lazy = (env != "0") and (env == "1" or big)
        self.lazy = bool(lazy) and ((self.dense_mode == 0 and self.n_seg == 1) or self.dense_mode == 1)
        self.lazy_state = (_dev_bytes(int(self.lib.anirec_train_lazy_bytes(self.rows)), dev) if self.lazy else None)
        self._stepper = None
        self.stream = torch.cuda.Stream(device=dev)
        self.user_idx = self.anime_idx = self.rating = self.sched = None
        self.n_steps = 0
        self._trainer = None
        self._desc = None
        self.set_head()

    # ---- views ---------------------------------------------------------------------
    # The kernels run on the engine's PRIVATE stream: a view handed out while steps are still in flight would be read
    # by torch (on its own stream) before they have finished.  Every public view therefore waits for the engine's
    # stream first; the kernels' own plumbing uses the raw `_W` / `_M` / `_V`.
    @property
    def W(self):
        self.stream.synchronize()
        return self._W

    @property
    def M(self):
        self.stream.synchronize()
        return self._M

    @property
    def V(self):
        self.stream.synchronize()
        return self._V

    @property
    def U(self):
        return self.W[: self.n_user_rows]

    @property
    def A(self):
        return self.W[self.n_user_rows:]

    # ---- state ---------------------------------------------------------------------
    def read_state(self) -> np.ndarray:
        self.stream.synchronize()
        return np.frombuffer(self.state_buf.cpu().numpy().tobytes(), dtype=_lib.STATE_DTYPE)[0].copy()

    def write_state(self, rec: np.ndarray) -> None:
        self.stream.synchronize()
        raw = np.frombuffer(np.asarray(rec, dtype=_lib.STATE_DTYPE).tobytes(), dtype=np.uint8).copy()
        self.state_buf.copy_(torch.from_numpy(raw).to(self.device))
        torch.cuda.synchronize(self.device)

    def set_head(self, w=1.0, b=0.0, gamma=1.0, beta=0.0, mov_mean=0.0, mov_var=1.0,
                 adam_m=None, adam_v=None):
        rec = np.zeros((), dtype=_lib.STATE_DTYPE)
        rec["w"], rec["b"], rec["gamma"], rec["beta"] = w, b, gamma, beta
        rec["mov_mean"], rec["mov_var"] = mov_mean, mov_var
        if adam_m is not None:
            rec["adam_m"] = adam_m
        if adam_v is not None:
            rec["adam_v"] = adam_v
        self.write_state(rec)

    def set_weights(self, U, A):
        """Load embedding tables (numpy or torch, fp32) and refresh the L2 partial sums."""
        U = torch.as_tensor(U, dtype=torch.float32)
        A = torch.as_tensor(A, dtype=torch.float32)
        assert U.shape == (self.n_user_rows, DIM) and A.shape == (self.n_anime_rows, DIM)
        self.stream.synchronize()
        self._W[: self.n_user_rows].copy_(U)
        self._W[self.n_user_rows:].copy_(A)
        torch.cuda.synchronize(self.device)
        self.init_reg()

    def reset_optimizer(self):
        self.stream.synchronize()
        self._M.zero_()
        self._V.zero_()
        torch.cuda.synchronize(self.device)

    def optimizer_state(self, iterations=0):
        """Keras-Adam slots of every trainable (the optimizer part of model.save, neural_network.py:220-221):
        first/second moments of both tables and of (w, b, gamma, beta), and the step counter."""
        rec = self.read_state()
        nu = self.n_user_rows
        return {"user_embedding/m": self.M[:nu].cpu().numpy(), "user_embedding/v": self.V[:nu].cpu().numpy(),
                "anime_embedding/m": self.M[nu:].cpu().numpy(), "anime_embedding/v": self.V[nu:].cpu().numpy(),
                "head/m": np.array(rec["adam_m"], np.float32), "head/v": np.array(rec["adam_v"], np.float32),
                "iterations": np.array([int(iterations)], np.int64)}

    # ---- descriptor ----------------------------------------------------------------
    def _build_desc(self):
        d = _lib.TrainDesc()
        d.n_user_rows, d.n_anime_rows = self.n_user_rows, self.n_anime_rows
        d.max_batch, d.arena_steps = self.max_batch, self.arena_steps
        d.dense_mode, d.n_seg, d.my_seg = self.dense_mode, self.n_seg, self.my_seg
        d.dense_rows = self.dense_rows
        d.adam_row_lo, d.adam_row_hi = self.adam_rows
        d.l2 = self.l2
        d.W, d.M, d.V = _lib.ptr(self._W), _lib.ptr(self._M), _lib.ptr(self._V)
        d.rowmap, d.state = _lib.ptr(self.rowmap), _lib.ptr(self.state_buf)
        d.user_idx, d.anime_idx = _lib.ptr(self.user_idx), _lib.ptr(self.anime_idx)
        d.rating, d.sched = _lib.ptr(self.rating), _lib.ptr(self.sched)
        d.n_steps = self.n_steps
        d.packets = _lib.ptr(self.packets)
        d.dense_grad = _lib.ptr(self.dense_grad)
        d.workspace, d.workspace_bytes = _lib.ptr(self.workspace), self.workspace.numel()
        d.lazy, d.lazy_state = int(self.lazy), _lib.ptr(self.lazy_state)
        self._desc = d
        if self._trainer is not None:
            _lib.check(self.lib.anirec_trainer_destroy(self._trainer), "anirec_trainer_destroy")
            self._trainer = None
        if self._stepper is not None:
            _lib.check(self.lib.anirec_dist_stepper_destroy(self._stepper), "anirec_dist_stepper_destroy")
            self._stepper = None
        return d

    @property
    def desc(self):
        return self._desc if self._desc is not None else self._build_desc()

    def _sp(self):
        return C.c_void_p(self.stream.cuda_stream)

    def init_reg(self):
        _lib.check(self.lib.anirec_train_init_reg(C.byref(self.desc), self._sp()), "anirec_train_init_reg")

    # ---- epoch data ----------------------------------------------------------------
    def set_epoch(self, user_idx, anime_idx, rating, starts, counts, alphas, global_counts=None):
        """Install one epoch: shuffled ratings (device int32/int32/fp32) + its step schedule."""
        self.stream.synchronize()
        dev = self.device
        self.user_idx = torch.as_tensor(user_idx, device=dev).to(torch.int32).contiguous()
        self.anime_idx = torch.as_tensor(anime_idx, device=dev).to(torch.int32).contiguous()
        self.rating = torch.as_tensor(rating, device=dev).to(torch.float32).contiguous()
        counts = np.asarray(counts, np.int32)
        if counts.size and int(counts.max()) > self.max_batch:
            raise ValueError("a step holds %d ratings > max_batch %d" % (counts.max(), self.max_batch))
        sched = np.zeros(len(counts), dtype=_lib.STEP_DTYPE)
        sched["start"], sched["count"], sched["alpha"] = starts, counts, alphas
        sched["global_count"] = counts if global_counts is None else global_counts
        raw = torch.from_numpy(np.frombuffer(sched.tobytes(), dtype=np.uint8).copy())
        self.sched = raw.to(dev)
        self.n_steps = len(counts)
        rec = self.read_state()
        rec["step_fwd"] = 0
        rec["step_bwd"] = 0
        self.write_state(rec)
        self._build_desc()
        # the L2 partial sums are double-buffered by step parity and the cursor restarts at 0: refresh both
        self.init_reg()

    def reset_metrics(self):
        rec = self.read_state()
        for k in ("loss_wsum", "se_sum", "n_seen", "val_bce_sum", "val_se_sum", "val_n", "bce_wsum",
                  "reg_user_wsum", "reg_anime_wsum"):
            rec[k] = 0.0
        self.write_state(rec)

    # ---- stages (unit-testable) -----------------------------------------------------
    def prep(self, first_step, n_steps):
        _lib.check(self.lib.anirec_train_prep(C.byref(self.desc), first_step, n_steps, self._sp()),
                   "anirec_train_prep")

    def fwd(self):
        _lib.check(self.lib.anirec_train_fwd(C.byref(self.desc), self._sp()), "anirec_train_fwd")

    def head(self):
        _lib.check(self.lib.anirec_train_head(C.byref(self.desc), self._sp()), "anirec_train_head")

    def bwd(self):
        _lib.check(self.lib.anirec_train_bwd(C.byref(self.desc), self._sp()), "anirec_train_bwd")

    def adam(self):
        _lib.check(self.lib.anirec_train_adam(C.byref(self.desc), self._sp()), "anirec_train_adam")

    def adam_users(self):
        _lib.check(self.lib.anirec_train_adam_part(C.byref(self.desc), 1, self._sp()), "anirec_train_adam_part")

    def adam_anime_finish(self):
        _lib.check(self.lib.anirec_train_adam_part(C.byref(self.desc), 2, self._sp()), "anirec_train_adam_part")

    def stage_ticks(self, enable=True, read=True):
        """Measurement hook (bench.py): in-kernel constant-clock stamps.  Returns {"fwd", "head", "bwd", "adam",
        "lazy_catchup", "lazy_adam", "lazy_flush", "lazy_reduce"} -> mean duration [us] of the launches made since the
        last call (None = not launched) and, under "launches", their counts; then arms / disarms.  Armed steps run
        eagerly and synchronise after every launch."""
        us = (C.c_float * 8)()
        nl = (C.c_int32 * 8)()
        _lib.check(self.lib.anirec_train_stage_ticks(C.byref(self.desc), int(bool(enable)), us, nl, self._sp()),
                   "anirec_train_stage_ticks")
        if not read:
            return None
        names = ("fwd", "head", "bwd", "adam", "lazy_catchup", "lazy_adam", "lazy_flush", "lazy_reduce")
        out = {k: (float(us[i]) if nl[i] else None) for i, k in enumerate(names)}
        out["launches"] = {k: int(nl[i]) for i, k in enumerate(names)}
        return out

    # ---- multi-GPU step halves: one C call each, the collectives go between them ------------
    def _get_stepper(self):
        if self._stepper is None:
            h = C.c_void_p()
            _lib.check(self.lib.anirec_dist_stepper_create(C.byref(self.desc), C.byref(h)),
                       "anirec_dist_stepper_create")
            self._stepper = h
        return self._stepper

    def stepper_begin(self, first_step, n_steps):
        """A run of n_steps steps driven through step_front / step_mid / step_back starts at first_step (the tables
        are current there); required with lazy user rows, a no-op otherwise."""
        _lib.check(self.lib.anirec_dist_stepper_begin(self._get_stepper(), int(first_step), int(n_steps), self._sp()),
                   "anirec_dist_stepper_begin")

    def stepper_block(self, n_steps):
        """The next n_steps steps have just been prepared (``prep``)."""
        _lib.check(self.lib.anirec_dist_stepper_block(self._get_stepper(), int(n_steps)), "anirec_dist_stepper_block")

    def step_front(self):
        _lib.check(self.lib.anirec_dist_step_front(self._get_stepper(), self._sp()), "anirec_dist_step_front")

    def step_mid(self):
        _lib.check(self.lib.anirec_dist_step_mid(self._get_stepper(), self._sp()), "anirec_dist_step_mid")

    def step_back(self):
        _lib.check(self.lib.anirec_dist_step_back(self._get_stepper(), self._sp()), "anirec_dist_step_back")

    # ---- the hot loop ---------------------------------------------------------------
    def run(self, n_steps=None, use_graph=True, first_step=None):
        """Run n_steps optimiser steps from the current cursor (prep + fwd/head/bwd/adam)."""
        if self.n_seg != 1 or self.dense_mode:
            raise _lib.AnirecError("TrainEngine.run is the single-GPU loop; use DistTrainEngine")
        if first_step is None:
            first_step = int(self.read_state()["step_fwd"])
        if n_steps is None:
            n_steps = self.n_steps - first_step
        if self._trainer is None:
            h = C.c_void_p()
            _lib.check(self.lib.anirec_trainer_create(C.byref(self.desc), C.byref(h)), "anirec_trainer_create")
            self._trainer = h
        _lib.check(self.lib.anirec_trainer_run(self._trainer, int(first_step), int(n_steps), int(use_graph),
                                               self._sp()), "anirec_trainer_run")
        return n_steps

    def eval_sums(self, user_idx, anime_idx, rating):
        """Validation pass (BN inference) over the given rows; returns the raw sums
        (val_bce_sum, val_se_sum, val_n) and the L2 sums of the current tables."""
        dev = self.device
        u = torch.as_tensor(user_idx, device=dev).to(torch.int32).contiguous()
        a = torch.as_tensor(anime_idx, device=dev).to(torch.int32).contiguous()
        t = torch.as_tensor(rating, device=dev).to(torch.float32).contiguous()
        rec = self.read_state()
        rec["val_bce_sum"] = rec["val_se_sum"] = rec["val_n"] = 0.0
        self.write_state(rec)
        self.init_reg()  # reg sums of the CURRENT weights
        _lib.check(self.lib.anirec_eval(C.byref(self.desc), _lib.ptr(u), _lib.ptr(a), _lib.ptr(t),
                                        int(u.numel()), self._sp()), "anirec_eval")
        rec = self.read_state()
        return {k: float(rec[k]) for k in ("val_bce_sum", "val_se_sum", "val_n", "reg_user_sumsq",
                                           "reg_anime_sumsq", "reg_sumsq")}

    def evaluate(self, user_idx, anime_idx, rating):
        """Validation pass. Returns (val_loss, val_mse) with Keras semantics (val_loss includes
        the whole-table L2 term, neural_network.py:216)."""
        r = self.eval_sums(user_idx, anime_idx, rating)
        n = max(r["val_n"], 1.0)
        val_loss = np.float32(r["val_bce_sum"] / n) + np.float32(self.l2) * np.float32(r["reg_sumsq"])
        return float(val_loss), float(r["val_se_sum"] / n)

    def epoch_metrics(self):
        rec = self.read_state()
        n = max(float(rec["n_seen"]), 1.0)
        return float(rec["loss_wsum"] / n), float(rec["se_sum"] / n)

    def synchronize(self):
        self.stream.synchronize()

    def close(self):
        if self._trainer is not None:
            self.stream.synchronize()
            self.lib.anirec_trainer_destroy(self._trainer)
            self._trainer = None
        if self._stepper is not None:
            self.stream.synchronize()
            torch.cuda.synchronize(self.device)
            self.lib.anirec_dist_stepper_destroy(self._stepper)
            self._stepper = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- debug/test helper: python mirror of csrc/anirec_train.hip::carve() ----------------
def _align(x, a=256):
    return (x + a - 1) // a * a


def workspace_layout(max_batch, arena_steps):
    cap = max_batch
    capC = (cap + cap // _lib.CHUNK + 2 + 3) & ~3
    off = 0
    lay = {"cap": cap, "capC": capC}
    # hpart, pub, regpart, P, S hold two copies (step parity 0 then 1)
    for name, nbytes in (("su", 4 * cap), ("sa", 4 * cap), ("dy", 4 * cap),
                         ("hpart", 2 * 4 * 8 * _lib.MAX_SEG * ((cap + 255) // 256)), ("pub", 2 * 64), ("sel", 16),
                         ("regpart", 4 * 4 * _lib.ADAM_BLOCKS),
                         ("P", 2 * 4 * 2 * capC * DIM), ("S", 2 * 4 * 2 * capC)):
        lay[name] = (off, nbytes)
        off += _align(nbytes)
    a1 = _align(4 * 2 * cap)
    lay["slot_bytes"] = _align(16) + 2 * a1 + _align(16 * 2 * capC)
    lay["arena"] = off
    lay["slot_sidx"] = 256
    lay["slot_oth"] = 256 + a1
    lay["slot_chunks"] = 256 + 2 * a1
    lay["ticks"] = (off + lay["slot_bytes"] * arena_steps, 8 * 8 * 2 * _lib.ADAM_BLOCKS)
    lay["lzpart"] = (lay["ticks"][0] + lay["ticks"][1], 4 * _lib.LAZY_WINDOW * 2 * _lib.ADAM_BLOCKS)
    lay["lzring"] = (lay["lzpart"][0] + lay["lzpart"][1], _align(4 * _lib.LAZY_WINDOW * 2))
    lay["total"] = lay["lzring"][0] + lay["lzring"][1]
    return lay


def read_ws(engine, name, dtype=np.float32):
    """Copy one named workspace array back to the host (tests only)."""
    lay = workspace_layout(engine.max_batch, engine.arena_steps)
    off, nbytes = lay[name]
    engine.synchronize()
    raw = engine.workspace[off:off + nbytes].cpu().numpy()
    return np.frombuffer(raw.tobytes(), dtype=dtype)


def read_slot(engine, step):
    """Sorted-batch + chunk tables of one prepared step (tests only)."""
    lay = workspace_layout(engine.max_batch, engine.arena_steps)
    cap, capC = lay["cap"], lay["capC"]
    base = lay["arena"] + lay["slot_bytes"] * (step % engine.arena_steps)
    engine.synchronize()
    raw = engine.workspace[base:base + lay["slot_bytes"]].cpu().numpy().tobytes()
    nch = np.frombuffer(raw[:8], dtype=np.int32).copy()
    sidx = np.frombuffer(raw[lay["slot_sidx"]:lay["slot_sidx"] + 8 * cap], dtype=np.int32).reshape(2, cap).copy()
    oth = np.frombuffer(raw[lay["slot_oth"]:lay["slot_oth"] + 8 * cap], dtype=np.int32).reshape(2, cap).copy()
    chunks = np.frombuffer(raw[lay["slot_chunks"]:lay["slot_chunks"] + 32 * capC], dtype=np.int32).reshape(2, capC, 4).copy()
    return nch, sidx, oth, chunks
