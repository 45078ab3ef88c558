"""Host logic of the training component vs the oracle and the reference's History CSV."""
import csv
import os

import numpy as np

from anime_recommendations_amd import schedule
from oracle import anirec_oracle as orc


def test_lrfn_matches_reference_history_and_oracle(golden_dir):
    with open(os.path.join(golden_dir, "anime_nn_history.csv")) as f:
        rows = list(csv.DictReader(f))
    for e, r in enumerate(rows):
        assert np.float32(schedule.lrfn(e)) == np.float32(float(r["lr"]))
    for e in range(40):
        for kw in ({}, dict(sustain_epochs=3), dict(rampup_epochs=2, exp_decay=0.5, min_lr=2e-6)):
            assert schedule.lrfn(e, **kw) == orc.lrfn(e, **kw)


def test_adam_alpha_bitwise_equals_oracle():
    for lr in (1e-5, 4.2e-5, 1e-3):
        for t in (1, 2, 10, 699, 12345, 218000):
            assert schedule.adam_alpha(lr, t) == orc.adam_alpha(lr, t)
        v = schedule.adam_alphas(lr, 5, 50)
        assert all(v[i] == orc.adam_alpha(lr, 5 + i) for i in range(50))
