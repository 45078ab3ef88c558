"""Host mirror of the reference's training driver (neural_network/neural_network.py:141-233):
``model.fit`` with LearningRateScheduler(lrfn), ModelCheckpoint(save_best_only on val_loss) and
EarlyStopping(patience=3, restore_best_weights=True), producing the Keras ``History`` columns
``loss, mse, val_loss, val_mse, lr``.

All arithmetic of a step runs in libanirec (HIP); this file only sequences epochs.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import numpy as np
import torch

from . import ops, schedule
from .data import RatingTable


@dataclass
class FitConfig:
    epochs: int = 20                 # config.yaml:68
    batch_size: int = 10_000         # config.yaml:59
    test_size: int = 10_000          # config.yaml:55
    embedding_size: int = 128        # config.yaml:63
    l2_reg_factor: float = 1e-4      # config.yaml:64
    start_lr: float = 1e-5           # implied by figure_file/anime_nn_history.csv (missing in YAML)
    min_lr: float = 1e-5
    max_lr: float = 5e-5
    rampup_epochs: int = 5
    sustain_epochs: int = 0
    exp_decay: float = 0.8
    patience: int = 3                # EarlyStopping(patience=3)  neural_network.py:198
    monitor: str = "val_loss"
    mode: str = "min"
    restore_best_weights: bool = True
    seed: int = 0                    # weight init + epoch shuffles (the reference is unseeded)
    verbose: int = 1
    use_graph: bool = True
    arena_steps: int = 64

    def lr(self, epoch):
        return schedule.lrfn(epoch, self.start_lr, self.max_lr, self.min_lr, self.rampup_epochs,
                             self.sustain_epochs, self.exp_decay)


@dataclass
class FitResult:
    history: dict
    U: np.ndarray
    A: np.ndarray
    head: dict
    best_U: np.ndarray = None
    best_A: np.ndarray = None
    best_head: dict = None
    best_epoch: int = -1
    stopped_epoch: int = -1
    optimizer: dict = field(default_factory=dict)
    step_loop_seconds: list = field(default_factory=list)   # per epoch: wall time of the step loop alone (synchronised)
    epoch_seconds: list = field(default_factory=list)       # per epoch: shuffle + steps + metrics + validation + snapshot


def init_weights(n_users, n_anime, dim=128, seed=0):
    """Keras initialisers at the reference's call sites: Embedding 'uniform' = U(-0.05, 0.05)
    (neural_network.py:75-85); Dense(1, he_normal) = truncated normal, stddev
    sqrt(2/fan_in)/0.87962566103423978 with fan_in = 1, cut at 2 stddev (neural_network.py:97)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    U = rng.uniform(-0.05, 0.05, (n_users, dim)).astype(np.float32)
    A = rng.uniform(-0.05, 0.05, (n_anime, dim)).astype(np.float32)
    std = np.sqrt(2.0 / 1.0) / 0.87962566103423978
    w = rng.normal(0.0, std)
    while abs(w) > 2 * std:
        w = rng.normal(0.0, std)
    return U, A, float(np.float32(w))


def head_of(rec):
    return {k: float(rec[k]) for k in ("w", "b", "gamma", "beta", "mov_mean", "mov_var")}


def _improved(cur, best, mode):
    return cur < best if mode == "min" else cur > best


def _column(col, rows, dtype, dev):
    """rows ``rows`` of a table column (NumPy on the host or a torch tensor already in HBM) as a device tensor"""
    col = col[rows]
    if isinstance(col, torch.Tensor):
        return col.to(device=dev, dtype=dtype).contiguous()
    np_dtype = {torch.int32: np.int32, torch.float32: np.float32}[dtype]
    return torch.as_tensor(np.asarray(col, np_dtype), device=dev)


def fit(table: RatingTable, cfg: FitConfig, engine=None, log=print, device="cuda:0") -> FitResult:
    """Train the embedding model on ``table`` (a ``data.RatingTable`` of NumPy columns or an
    ``ingest.EncodedRatings`` whose columns already live in HBM); returns History + last and best weights."""
    if cfg.embedding_size != 128:
        raise ValueError("libanirec kernels are specialised for embedding_size 128 (config.yaml:63)")
    tr, te = table.split(cfg.test_size)
    n_train = tr.stop - tr.start
    if engine is None:
        from .engine import TrainEngine
        engine = TrainEngine(table.n_users, table.n_anime, max_batch=min(cfg.batch_size, n_train),
                             l2=cfg.l2_reg_factor, arena_steps=cfg.arena_steps, device=device)
    dev = engine.device
    U0, A0, w0 = init_weights(table.n_users, table.n_anime, 128, cfg.seed)
    engine.set_head(w=w0)
    engine.set_weights(U0, A0)
    engine.reset_optimizer()

    ui, ai, rt = (_column(c, tr, dt, dev) for c, dt in ((table.user, torch.int32), (table.anime, torch.int32),
                                                        (table.rating, torch.float32)))
    vu, va, vt = (_column(c, te, dt, dev) for c, dt in ((table.user, torch.int32), (table.anime, torch.int32),
                                                        (table.rating, torch.float32)))

    B = min(cfg.batch_size, n_train)
    # multi-GPU (dist.DistTrainEngine): each rank takes B ratings of a global batch of G*B
    Bg = int(getattr(engine, "global_batch", B))
    starts = np.arange(0, n_train, Bg)
    counts = np.minimum(Bg, n_train - starts)
    n_steps = len(starts)
    multi = hasattr(engine, "set_epoch_global")
    gen = torch.Generator(device=dev)
    hist = {"loss": [], "mse": [], "val_loss": [], "val_mse": [], "lr": []}
    best = np.inf if cfg.mode == "min" else -np.inf
    best_w = None
    best_epoch, stopped, wait = -1, -1, 0
    t_global = 0
    loop_s, epoch_s = [], []
    for epoch in range(cfg.epochs):
        t_epoch = time.perf_counter()
        lr = cfg.lr(epoch)
        gen.manual_seed(cfg.seed * 1_000_003 + epoch)
        perm = torch.randperm(n_train, generator=gen, device=dev)     # model.fit(shuffle=True)
        alphas = schedule.adam_alphas(lr, t_global + 1, n_steps)
        if multi:
            import torch.distributed as dist
            if dist.is_initialized() and dist.get_world_size() > 1:
                dist.broadcast(perm, src=0)                           # one shuffle for all ranks
            engine.set_epoch_global(ui, ai, rt, perm, alphas)
        else:
            eu, ea, et = ops.gather_ratings(ui, ai, rt, perm)
            engine.set_epoch(eu, ea, et, starts, counts, alphas)
        engine.reset_metrics()
        engine.synchronize()
        t_loop = time.perf_counter()
        engine.run(n_steps, use_graph=cfg.use_graph)
        engine.synchronize()
        loop_s.append(time.perf_counter() - t_loop)
        t_global += n_steps
        loss, mse = engine.epoch_metrics()
        val_loss, val_mse = engine.evaluate(vu, va, vt)
        for k, v in (("loss", loss), ("mse", mse), ("val_loss", val_loss), ("val_mse", val_mse),
                     ("lr", float(np.float32(lr)))):
            hist[k].append(v)
        if cfg.verbose:
            log("Epoch %d/%d - loss: %.4f - mse: %.4f - val_loss: %.4f - val_mse: %.4f - lr: %.4g"
                % (epoch + 1, cfg.epochs, loss, mse, val_loss, val_mse, lr))
        cur = hist[cfg.monitor][-1]
        if _improved(cur, best, cfg.mode):                             # ModelCheckpoint / best_weights
            best, best_epoch, wait = cur, epoch, 0
            # snapshot on the device (a 188 MB table copies in ~0.1 ms there, ~60 ms through the host);
            # it is brought to the host once, after the last epoch
            engine.synchronize()
            best_w = (engine.U.clone(), engine.A.clone(), head_of(engine.read_state()))
            if torch.device(dev).type == "cuda":
                torch.cuda.synchronize(dev)   # the clones ran on torch's stream: finish before the next epoch writes W
        else:
            wait += 1
        engine.synchronize()
        epoch_s.append(time.perf_counter() - t_epoch)
        if wait >= cfg.patience and epoch > 0:                        # EarlyStopping
            stopped = epoch
            break
    engine.synchronize()
    rec = engine.read_state()
    res = FitResult(history=hist, U=engine.U.cpu().numpy().copy(), A=engine.A.cpu().numpy().copy(),
                    head=head_of(rec), best_epoch=best_epoch, stopped_epoch=stopped, step_loop_seconds=loop_s,
                    epoch_seconds=epoch_s)
    if hasattr(engine, "optimizer_state"):       # Adam m, v and the step count of the LAST epoch (model.save)
        res.optimizer = engine.optimizer_state(iterations=t_global)
    if best_w is not None:
        best_w = (best_w[0].cpu().numpy(), best_w[1].cpu().numpy(), best_w[2])
        res.best_U, res.best_A, res.best_head = best_w
        if stopped >= 0 and cfg.restore_best_weights:
            res.U, res.A, res.head = best_w
    return res


def history_frame(history):
    """pandas frame with the reference's History CSV layout (`,loss,mse,val_loss,val_mse,lr`)."""
    import pandas as pd
    return pd.DataFrame({k: history[k] for k in ("loss", "mse", "val_loss", "val_mse", "lr")})
