"""Learning-rate schedule and Adam step size of the training component (host logic).

Mirrors ``lrfn`` (reference neural_network/neural_network.py:109-125) driven by
``LearningRateScheduler`` (:184-186), and the bias-corrected step size of the Keras-2.12
Adam the reference compiles with (``optimizer='Adam'``, :104).
"""
from __future__ import annotations

import numpy as np

ADAM_B1 = 0.9
ADAM_B2 = 0.999


def lrfn(epoch, start_lr=1e-5, max_lr=5e-5, min_lr=1e-5, rampup_epochs=5, sustain_epochs=0,
         exp_decay=0.8):
    """Learning rate of ``epoch`` (0-based): linear ramp start->max over ``rampup_epochs``,
    hold for ``sustain_epochs``, then exponential decay towards ``min_lr``."""
    start_lr, max_lr, min_lr, exp_decay = float(start_lr), float(max_lr), float(min_lr), float(exp_decay)
    rampup_epochs, sustain_epochs = int(rampup_epochs), int(sustain_epochs)
    if epoch < rampup_epochs:
        return (max_lr - start_lr) / rampup_epochs * epoch + start_lr
    if epoch < rampup_epochs + sustain_epochs:
        return max_lr
    return (max_lr - min_lr) * exp_decay ** (epoch - rampup_epochs - sustain_epochs) + min_lr


def adam_alpha(lr, t):
    """lr * sqrt(1 - b2^t) / (1 - b1^t) in fp32, t = 1-based optimiser iteration."""
    f = np.float32
    lr, ts = f(lr), f(t)
    b1p = np.power(f(ADAM_B1), ts, dtype=f)
    b2p = np.power(f(ADAM_B2), ts, dtype=f)
    return f(lr * np.sqrt(f(1) - b2p, dtype=f) / (f(1) - b1p))


def adam_alphas(lr, t_first, n):
    """Vector of step sizes for iterations t_first .. t_first+n-1 (fp32)."""
    f = np.float32
    ts = np.arange(t_first, t_first + n, dtype=f)
    b1p = np.power(f(ADAM_B1), ts, dtype=f)
    b2p = np.power(f(ADAM_B2), ts, dtype=f)
    return (f(lr) * np.sqrt(f(1) - b2p, dtype=f) / (f(1) - b1p)).astype(f)
