"""ctypes binding of libmfcd_hip.so (C-ABI declared in include/mfcd.h).

There is no fallback: if the HIP library is not built, or a call is made without a GPU tensor,
this module raises.  PyTorch is used only to own device memory and streams.
"""
import ctypes
import os

import torch

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("MFCD_LIB") or os.path.join(_PKG_DIR, "libmfcd_hip.so")  # MFCD_LIB: diagnostic builds (tools/)

_vp, _i32, _i64, _dbl, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/mfcd.h one to one
SIGNATURES = {
    "mfcd_abi_version": (_i32, []),
    "mfcd_error_string": (ctypes.c_char_p, [_i32]),
    "mfcd_check_samples": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "mfcd_eval_batches": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "mfcd_train_workspace_bytes": (_sz, [_i64, _i32, _i32, _i32, _i32]),
    "mfcd_train_workspace_init": (_i32, [_vp, _sz, _i64, _i32, _i32, _i32, _i32, _vp]),
    "mfcd_train_workspace_release": (_i32, [_vp]),
    "mfcd_set_train_path": (_i32, [_i32]),
    "mfcd_set_resident_math": (_i32, [_i32]),
    "mfcd_set_tuning": (_i32, [_i32, _i64]),
    "mfcd_train_plan_query": (_i32, [_i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mfcd_train_steps": (_i32, [_vp] * 7 + [_i64, _i32, _i64, _i32, _i32, _i32] + [_dbl] * 5 + [_vp, _vp, _sz, _vp]),
    "mfcd_train_steps_bf16": (_i32, [_vp] * 7 + [_i64, _i32, _i64, _i32, _i32, _i32] + [_dbl] * 5 + [_vp, _vp, _sz, _vp]),
    "mfcd_eval_batches_bf16": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "mfcd_train_steps_timed": (_i32, [_vp] * 7 + [_i64, _i32, _i64, _i32, _i32, _i32] + [_dbl] * 5 +
                               [_vp, _vp, _sz, _vp, _vp]),
    "mfcd_train_call_prepare": (_i32, [_vp] * 6 + [_i32] * 5 + [_dbl] * 5 + [_vp, _sz, ctypes.POINTER(ctypes.c_void_p)]),
    "mfcd_train_call_run": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "mfcd_train_call_stage": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "mfcd_train_call_release": (_i32, [_vp]),
    "mfcd_batch_coefficients": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "mfcd_apply_step": (_i32, [_vp] * 8 + [_i32, _i64, _i32, _i32, _i32] + [_dbl] * 5 + [_vp, _sz, _vp]),
    "mfcd_dense_grad": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "mfcd_dense_grad_from_coefficients": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "mfcd_adam_dense": (_i32, [_vp] * 8 + [_i64, _i32, _i32, _i32] + [_dbl] * 5 + [_vp]),
    "mfcd_dp_unique_id": (_i32, [_vp, _sz]),
    "mfcd_dp_comm_create": (_i32, [_vp, _sz, _i32, _i32, ctypes.POINTER(ctypes.c_void_p)]),
    "mfcd_dp_comm_destroy": (_i32, [_vp]),
    "mfcd_dp_workspace_bytes": (_sz, [_i64, _i32, _i32, _i32, _i32, _i32]),
    "mfcd_dp_train_steps": (_i32, [_vp] * 7 + [_i64, _i32, _i32, _i32, _i64, _i32, _i32, _i32] + [_dbl] * 5 +
                            [_vp, _vp, _sz, _vp, _vp]),
    "mfcd_dp_train_steps_bf16": (_i32, [_vp] * 7 + [_i64, _i32, _i32, _i32, _i64, _i32, _i32, _i32] + [_dbl] * 5 +
                            [_vp, _vp, _sz, _vp, _vp]),
    "mfcd_shard_rows": (_i32, [_i32, _i32, _i32, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "mfcd_shard_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "mfcd_shard_pack": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "mfcd_shard_collisions": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "mfcd_shard_pack_ahead": (_i32, [_vp] * 7 + [_i32, _i32, _i64, _i32, _i32, _i32, _i32, _i32] + [_dbl] * 5 + [_vp, _vp]),
    "mfcd_shard_apply": (_i32, [_vp] * 7 + [_i32, _i32, _vp, _i64, _i32, _i32, _i32, _i32, _i32] + [_dbl] * 5 + [_vp, _vp]),
    "mfcd_shard_train_steps": (_i32, [_vp] * 7 + [_i64, _i32, _i32, _i32, _i64, _i32, _i32, _i32] + [_dbl] * 5 +
                               [_vp, _vp, _sz, _vp, _vp]),
    "mfcd_shard_train_steps_bf16": (_i32, [_vp] * 7 + [_i64, _i32, _i32, _i32, _i64, _i32, _i32, _i32] + [_dbl] * 5 +
                               [_vp, _vp, _sz, _vp, _vp]),
    "mfcd_uvt_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "mfcd_uvt_stats": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _dbl, _vp, _vp, _vp, _sz, _vp]),
    "mfcd_uvt_stats_select": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _dbl, _i32, _vp, _vp, _vp, _sz, _vp]),
    "mfcd_uvt_slab_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "mfcd_uvt_stats_slab": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _dbl, _i32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "mfcd_uvt_rows": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "mfcd_generate_labels": (_i32, [_vp, _i64, _vp, _i32, _i32, _vp, _vp, _i32, _dbl, _i32, _i32, ctypes.c_uint64, _vp,
                                     _vp]),
    "mfcd_train_big_workspace_bytes": (_sz, [_i64, _i32]),
    "mfcd_train_steps_big": (_i32, [_vp] * 7 + [_i64, _i32, _i64, _i32, _i32, _i32] + [_dbl] * 5 + [_vp, _vp, _sz, _vp]),
    "mfcd_train_big_status": (_i32, [_vp, ctypes.POINTER(ctypes.c_int), _vp]),
    "mfcd_train_big_slots": (_i32, []),
    "mfcd_train_big_check": (_i32, [_vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "mfcd_sample_workspace_bytes": (_sz, [_i64, _i64]),
    "mfcd_sample_triplets": (_i32, [_vp, _vp, _i64, _i64, _i64, ctypes.c_uint64, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mfcd_spearman_max_columns": (_i32, []),
    "mfcd_spearman_rows": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _vp, _vp]),
    "mfcd_spearman_long_workspace_bytes": (_sz, [_i32, _i32]),
    "mfcd_spearman_rows_long": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _vp, _vp, _sz, _vp]),
}

TUNE_KEYS = {"resident_q": 1, "resident_wpc": 2, "resident_lookahead": 3, "resident_lds_pad": 4,
             "resident_spin_limit": 5, "short_call_steps": 6, "uvt_wpe128": 7, "stream_chunks": 8,
             "uvt_target_wgs": 9, "uvt_min_stages": 10, "uvt_split": 11, "rank_sort": 12, "shard_pipeline": 13}


class TrainPlan(ctypes.Structure):
    """mfcd_train_plan of include/mfcd.h."""
    _fields_ = [(k, ctypes.c_int32) for k in ("form", "resident_q", "resident_waves", "resident_blocks",
                                               "resident_lookahead", "fast_math", "streaming_vec",
                                               "streaming_chunks", "streaming_blocks", "device_cus")] + \
               [("reserved", ctypes.c_int32 * 6)]


class Sampler(ctypes.Structure):
    """mfcd_sampler of include/mfcd.h."""
    _fields_ = [("law", ctypes.c_int32), ("n", ctypes.c_int32), ("m", ctypes.c_int32), ("pair_rule", ctypes.c_int32),
                ("cdf", _vp), ("list_i", _vp), ("list_j", _vp), ("k", ctypes.c_int32),
                ("list_row_stride", ctypes.c_int32), ("users", _vp), ("n_users", ctypes.c_int32),
                ("use_margin", ctypes.c_int32), ("margin", _dbl), ("X", _vp), ("A", _vp), ("B", _vp),
                ("dx", ctypes.c_int32), ("reserved", ctypes.c_int32)]


_lib = None


class MfcdError(RuntimeError):
    pass


def load():
    """Load the shared library and bind every declared symbol (no GPU needed for this)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MfcdError(
                f"{LIB_PATH} is missing: the HIP hot path has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or make -C csrc). "
                "There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export it
            fn.restype, fn.argtypes = res, args
        if lib.mfcd_abi_version() != 4:
            raise MfcdError("libmfcd_hip.so ABI version mismatch")
        _lib = lib
    return _lib


def check(code):
    if code != 0:
        raise MfcdError(load().mfcd_error_string(code).decode())


def ptr(t):
    """Raw device pointer of a CUDA(HIP) tensor; None -> NULL."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MfcdError("the HIP hot path needs tensors on a GPU device (got a CPU tensor)")
    if not t.is_contiguous():
        raise MfcdError("the HIP hot path needs contiguous tensors")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)   # the handle without building a Stream object


def stream_ptr(device=None):
    if _raw_stream is not None and isinstance(device, torch.device) and device.index is not None:
        return _raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream
