"""ctypes binding of libanirec.so (include/anirec.h).

The product path has NO fallback: if the HIP library is missing or a call fails,
this module raises.  Nothing here imports the oracle.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ANIREC_LIB_PATH") or os.path.join(_PKG, "libanirec.so")   # override: A/B of two builds on one box

DIM = 128
MAX_BATCH = 16384
CHUNK = 32
ADAM_BLOCKS = 8192
MAX_TOPK = 128
MAX_SEG = 16
ABI_VERSION = 4
TOPK_MAX_BATCHES = 64
RCCL_ID_BYTES = 128
LAZY_WINDOW = 8


class AnirecError(RuntimeError):
    pass


class Step(C.Structure):
    _fields_ = [("start", C.c_int32), ("count", C.c_int32), ("alpha", C.c_float),
                ("global_count", C.c_int32)]


# numpy mirror of anirec_step / anirec_state (device buffers are viewed through these)
STEP_DTYPE = np.dtype([("start", "<i4"), ("count", "<i4"), ("alpha", "<f4"), ("global_count", "<i4")])
STATE_DTYPE = np.dtype([
    ("w", "<f4"), ("b", "<f4"), ("gamma", "<f4"), ("beta", "<f4"),
    ("adam_m", "<f4", (4,)), ("adam_v", "<f4", (4,)),
    ("mov_mean", "<f4"), ("mov_var", "<f4"), ("reg_sumsq", "<f4"),
    ("bn_mu", "<f4"), ("bn_var", "<f4"), ("last_loss", "<f4"), ("last_mse", "<f4"),
    ("step_fwd", "<i4"), ("step_bwd", "<i4"), ("pad0", "<i4"),
    ("loss_wsum", "<f8"), ("se_sum", "<f8"), ("n_seen", "<f8"),
    ("val_bce_sum", "<f8"), ("val_se_sum", "<f8"), ("val_n", "<f8"),
    ("bce_wsum", "<f8"), ("reg_user_wsum", "<f8"), ("reg_anime_wsum", "<f8"),
    ("reg_user_sumsq", "<f4"), ("reg_anime_sumsq", "<f4"),
], align=True)
assert STEP_DTYPE.itemsize == 16
assert STATE_DTYPE.itemsize == 168, STATE_DTYPE.itemsize


class TrainDesc(C.Structure):
    _fields_ = [
        ("n_user_rows", C.c_int32), ("n_anime_rows", C.c_int32), ("max_batch", C.c_int32),
        ("arena_steps", C.c_int32), ("dense_mode", C.c_int32), ("n_seg", C.c_int32),
        ("my_seg", C.c_int32), ("dense_rows", C.c_int32), ("l2", C.c_float), ("adam_row_lo", C.c_int32),
        ("adam_row_hi", C.c_int32), ("lazy", C.c_int32),
        ("W", C.c_void_p), ("M", C.c_void_p), ("V", C.c_void_p), ("rowmap", C.c_void_p),
        ("state", C.c_void_p), ("user_idx", C.c_void_p), ("anime_idx", C.c_void_p),
        ("rating", C.c_void_p), ("sched", C.c_void_p), ("n_steps", C.c_int32), ("pad2", C.c_int32),
        ("packets", C.c_void_p), ("dense_grad", C.c_void_p), ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_size_t), ("lazy_state", C.c_void_p),
    ]


class Head(C.Structure):
    _fields_ = [("w", C.c_float), ("b", C.c_float), ("gamma", C.c_float), ("beta", C.c_float),
                ("mov_mean", C.c_float), ("mov_var", C.c_float)]


class IngestOpts(C.Structure):
    _fields_ = [("num_reviews", C.c_int32), ("drop_unwatched", C.c_int32), ("drop_plan", C.c_int32),
                ("drop_half_watched", C.c_int32), ("user_id_bound", C.c_int32), ("anime_id_bound", C.c_int32)]


_vp, _i32, _sz, _f32, _i64 = C.c_void_p, C.c_int32, C.c_size_t, C.c_float, C.c_int64
_DP = C.POINTER(TrainDesc)

# name -> (restype, argtypes); must list every function include/anirec.h declares
PROTOTYPES = {
    "anirec_abi_version": (C.c_int, []),
    "anirec_status_string": (C.c_char_p, [C.c_int]),
    "anirec_device_name": (C.c_int, [C.c_char_p, _sz]),
    "anirec_packet_floats": (_sz, [_i32]),
    "anirec_train_workspace_bytes": (_sz, [_i32, _i32]),
    "anirec_train_lazy_bytes": (_sz, [_i32]),
    "anirec_train_init_reg": (C.c_int, [_DP, _vp]),
    "anirec_train_prep": (C.c_int, [_DP, _i32, _i32, _vp]),
    "anirec_train_fwd": (C.c_int, [_DP, _vp]),
    "anirec_train_head": (C.c_int, [_DP, _vp]),
    "anirec_train_bwd": (C.c_int, [_DP, _vp]),
    "anirec_train_adam": (C.c_int, [_DP, _vp]),
    "anirec_train_adam_part": (C.c_int, [_DP, _i32, _vp]),
    "anirec_train_stage_ticks": (C.c_int, [_DP, _i32, C.POINTER(C.c_float), C.POINTER(C.c_int32), _vp]),
    "anirec_dist_stepper_create": (C.c_int, [_DP, C.POINTER(_vp)]),
    "anirec_dist_stepper_destroy": (C.c_int, [_vp]),
    "anirec_dist_step_front": (C.c_int, [_vp, _vp]),
    "anirec_dist_step_mid": (C.c_int, [_vp, _vp]),
    "anirec_dist_step_back": (C.c_int, [_vp, _vp]),
    "anirec_dist_stepper_begin": (C.c_int, [_vp, _i32, _i32, _vp]),
    "anirec_dist_stepper_block": (C.c_int, [_vp, _i32]),
    "anirec_rccl_load": (C.c_int, [C.c_char_p]),
    "anirec_rccl_unique_id": (C.c_int, [C.c_char_p]),
    "anirec_dist_comm_create": (C.c_int, [C.c_char_p, _i32, _i32, C.POINTER(_vp)]),
    "anirec_dist_comm_destroy": (C.c_int, [_vp]),
    "anirec_dist_run": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "anirec_trainer_create": (C.c_int, [_DP, C.POINTER(_vp)]),
    "anirec_trainer_destroy": (C.c_int, [_vp]),
    "anirec_trainer_run": (C.c_int, [_vp, _i32, _i32, _i32, _vp]),
    "anirec_eval": (C.c_int, [_DP, _vp, _vp, _vp, _i32, _vp]),
    "anirec_adam_flat": (C.c_int, [_vp, _vp, _vp, _vp, _sz, _f32, _vp]),
    "anirec_selftest_lazy_math": (C.c_int, [C.c_uint64, _vp, _vp]),
    "anirec_gather_ratings": (C.c_int, [_vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "anirec_rownorm": (C.c_int, [_vp, _i32, _vp, _vp]),
    "anirec_cosine_scores": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "anirec_topk_workspace_bytes": (_sz, [_i32, _i32]),
    "anirec_cosine_topk": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "anirec_topk_mfma_workspace_bytes": (_sz, [_i32, _i32]),
    "anirec_cosine_topk_mfma": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "anirec_cosine_topk_mfma_prior": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, C.c_float, _vp, _vp, _vp, _vp, _sz,
                                                _vp]),
    "anirec_cosine_topk_job_plan": (C.c_int, [_i32, _i32, _i32, _i32, _i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                              C.POINTER(C.c_int32)]),
    "anirec_cosine_topk_job_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "anirec_cosine_topk_allpairs_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "anirec_cosine_topk_allpairs_plan": (C.c_int, [_i32, _i32, _i32, _i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                                   C.POINTER(C.c_int32)]),
    "anirec_cosine_topk_job": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, C.c_float, C.POINTER(C.c_int32), _i32,
                                         _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "anirec_topk_mfma_timing": (C.c_int, [_i32, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "anirec_predict_pairs": (C.c_int, [_vp, _vp, _vp, _vp, _i32, C.POINTER(Head), _vp, _vp]),
    "anirec_predict_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "anirec_predict_grid": (C.c_int, [_vp, _vp, _i32, _vp, _i32, C.POINTER(Head), _vp, _vp, _sz, _vp]),
    "anirec_predict_mfma_workspace_bytes": (_sz, [_i32, _i32]),
    "anirec_predict_grid_mfma": (C.c_int, [_vp, _vp, _i32, _vp, _i32, C.POINTER(Head), _vp, _vp, _sz, _vp]),
    "anirec_predict_topk": (C.c_int, [_vp, _vp, _i32, _vp, _i32, C.POINTER(Head), _vp, _i32, _vp,
                                      _vp, _vp, _sz, _vp]),
    "anirec_predict_topk_mfma_workspace_bytes": (_sz, [_i32, _i32]),
    "anirec_predict_topk_mfma": (C.c_int, [_vp, _vp, _i32, _vp, _i32, C.POINTER(Head), _vp, _i32, _vp,
                                           _vp, _vp, _vp, _sz, _vp]),
    "anirec_fav_workspace_bytes": (_sz, [_i64, _i32]),
    "anirec_user_favourites": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, C.c_double, _vp, _vp, _vp, _vp, _sz, _vp]),
    "anirec_user_recs": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "anirec_ingest_id_max": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "anirec_ingest_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "anirec_ingest_preprocess": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, C.POINTER(IngestOpts), _vp, _vp, _vp,
                                           _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "anirec_ingest_half_columns": (C.c_int, [_vp, _vp, _i64, C.POINTER(IngestOpts), _vp, _vp, _vp, _sz, _vp]),
    "anirec_ingest_encode_workspace_bytes": (_sz, [_i64, _i32]),
    "anirec_ingest_encode": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
}

_lib = None


def load() -> C.CDLL:
    """Load libanirec.so (built in-tree by anime_recommendations_amd.build).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AnirecError(
            "libanirec.so not found at %s — run `python -m anime_recommendations_amd.build` "
            "(needs hipcc); there is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    v = lib.anirec_abi_version()
    if v != ABI_VERSION:
        raise AnirecError("libanirec ABI %d != binding ABI %d" % (v, ABI_VERSION))
    _lib = lib
    return lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = load().anirec_status_string(status).decode()
        raise AnirecError("%s failed: %s (status %d)" % (what or "libanirec call", msg, status))


def ptr(t) -> int:
    """Device pointer of a torch tensor (must be contiguous); None -> NULL."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise AnirecError("non-contiguous tensor passed to libanirec")
    return t.data_ptr()
