"""Model container: the tensors the reference keeps in ``wandb_anime_nn.h5`` under Keras layer
names (``user_embedding`` / ``anime_embedding``: config.yaml:85-86, read back by
similar_anime.py:155,164) stored as safetensors (h5py/TensorFlow are not installable here;
SURVEY.md §8(f)-1 lists .h5 interop as a later row).

Tensor names:  <ID_emb_name>/embeddings, <anime_emb_name>/embeddings, dense/kernel, dense/bias,
batch_normalization/{gamma,beta,moving_mean,moving_variance}; optimiser slots under adam/.
"""
from __future__ import annotations

import json

import numpy as np
from safetensors.numpy import load_file, save_file

HEAD_KEYS = ("w", "b", "gamma", "beta", "mov_mean", "mov_var")


def save_model(path, U, A, head, user_ids=None, anime_ids=None, user_name="user_embedding",
               anime_name="anime_embedding", optimizer=None, extra=None):
    t = {
        user_name + "/embeddings": np.ascontiguousarray(U, np.float32),
        anime_name + "/embeddings": np.ascontiguousarray(A, np.float32),
        "dense/kernel": np.array([[head["w"]]], np.float32),
        "dense/bias": np.array([head["b"]], np.float32),
        "batch_normalization/gamma": np.array([head["gamma"]], np.float32),
        "batch_normalization/beta": np.array([head["beta"]], np.float32),
        "batch_normalization/moving_mean": np.array([head["mov_mean"]], np.float32),
        "batch_normalization/moving_variance": np.array([head["mov_var"]], np.float32),
    }
    if user_ids is not None:
        t["index/user_ids"] = np.ascontiguousarray(user_ids, np.int64)
    if anime_ids is not None:
        t["index/anime_ids"] = np.ascontiguousarray(anime_ids, np.int64)
    for k, v in (optimizer or {}).items():
        t["adam/" + k] = np.ascontiguousarray(v)
    meta = {"format": "anime_recommendations_amd/1", "user_layer": user_name, "anime_layer": anime_name}
    meta.update({k: json.dumps(v) for k, v in (extra or {}).items()})
    save_file(t, path, metadata=meta)
    return path


def load_model(path, user_name="user_embedding", anime_name="anime_embedding"):
    """Returns dict(U, A, head, user_ids, anime_ids, optimizer)."""
    t = load_file(path)
    ukey, akey = user_name + "/embeddings", anime_name + "/embeddings"
    if ukey not in t or akey not in t:
        raise KeyError("model file %s has no layers %r / %r (has %s)" % (path, user_name, anime_name, sorted(t)))
    head = {"w": float(t["dense/kernel"].reshape(-1)[0]), "b": float(t["dense/bias"][0]),
            "gamma": float(t["batch_normalization/gamma"][0]), "beta": float(t["batch_normalization/beta"][0]),
            "mov_mean": float(t["batch_normalization/moving_mean"][0]),
            "mov_var": float(t["batch_normalization/moving_variance"][0])}
    return {"U": t[ukey], "A": t[akey], "head": head, "user_ids": t.get("index/user_ids"),
            "anime_ids": t.get("index/anime_ids"),
            "optimizer": {k[5:]: v for k, v in t.items() if k.startswith("adam/")}}
