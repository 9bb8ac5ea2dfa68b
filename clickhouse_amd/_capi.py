"""ctypes binding of libchgpu.so — exactly the symbols include/chgpu.h declares.

The product path has NO CPU fallback: if the HIP library is missing or fails to load, importing raises.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CHGPU_LIB", os.path.join(_HERE, "libchgpu.so"))  # CHGPU_LIB: developer override for A/B builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "chgpu.h")

# enums of include/chgpu.h
OK = 0
ERR_SIZES_MISMATCH, ERR_NOT_IMPLEMENTED, ERR_OOM, ERR_LOGICAL, ERR_BAD_ARGUMENTS, ERR_DEVICE, ERR_TOO_MANY_ROWS = -1, -2, -3, -4, -5, -6, -7
I64, U32, U64, F64, U8, I32, U16, I16, I8, F32 = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
EQ, NE, LT, GT, LE, GE = 0, 1, 2, 3, 4, 5
AGG_COUNT, AGG_SUM, AGG_AVG, AGG_MIN, AGG_MAX, AGG_ANY = 0, 1, 2, 3, 4, 5
ASOF_LESS, ASOF_GREATER, ASOF_LESS_OR_EQUALS, ASOF_GREATER_OR_EQUALS = 1, 2, 3, 4
JOIN_INNER, JOIN_LEFT, JOIN_RIGHT, JOIN_FULL = 0, 1, 2, 3
STRICT_ANY, STRICT_ALL, STRICT_SEMI, STRICT_ANTI = 0, 1, 2, 3
N_COUNTERS = 8
VAL_COL, VAL_MUL, VAL_PLUS, VAL_MINUS = 0, 1, 2, 3

_vp, _i, _u32, _u64, _i64 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_int64
_pp = C.POINTER(C.c_void_p)
_pu64 = C.POINTER(C.c_uint64)

SIGNATURES = {
    "chgpu_abi_version": (_i, []),
    "chgpu_last_error": (C.c_char_p, []),
    "chgpu_ctx_create": (_i, [_i, _vp, _pp]),
    "chgpu_ctx_destroy": (_i, [_vp]),
    "chgpu_ctx_synchronize": (_i, [_vp]),
    "chgpu_ctx_trim": (_i, [_vp]),
    "chgpu_ctx_set_option": (_i, [_vp, C.c_char_p, _i64]),
    "chgpu_ctx_counters": (_i, [_vp, _pu64]),
    "chgpu_timer_start": (_i, [_vp]),
    "chgpu_timer_stop_ms": (_i, [_vp, C.POINTER(C.c_double)]),
    "chgpu_col_upload": (_i, [_vp, _i, _vp, _u64, _pp]),
    "chgpu_col_alloc": (_i, [_vp, _i, _u64, _pp]),
    "chgpu_host_alloc": (_i, [C.c_size_t, _pp]),
    "chgpu_host_free": (_i, [_vp]),
    "chgpu_col_upload_async": (_i, [_vp, _i, _vp, _u64, _pp, _pu64]),
    "chgpu_upload_wait": (_i, [_vp, _u64]),
    "chgpu_col_wrap": (_i, [_vp, _i, _vp, _u64, _pp]),
    "chgpu_col_slice": (_i, [_vp, _vp, _u64, _u64, _pp]),
    "chgpu_col_concat": (_i, [_vp, _u32, _pp, _pp]),
    "chgpu_col_download": (_i, [_vp, _vp, _vp, _u64]),
    "chgpu_col_rows": (_u64, [_vp]),
    "chgpu_col_type": (_i, [_vp]),
    "chgpu_col_device_ptr": (_vp, [_vp]),
    "chgpu_col_free": (_i, [_vp]),
    "chgpu_decompress_frames": (_i, [_vp, _vp, _u32, _pu64, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(_u32), _pp]),
    "chgpu_col_from_bytes": (_i, [_vp, _vp, _u64, _i, _u64, _pp]),
    "chgpu_cmp_const": (_i, [_vp, _vp, _i, _i, _vp, _pp]),
    "chgpu_count_bytes_in_filter": (_i, [_vp, _vp, _pu64]),
    "chgpu_filter": (_i, [_vp, _vp, _vp, _i64, _pp, _pu64]),
    "chgpu_filter_columns": (_i, [_vp, _u32, _vp, _vp, _i64, _vp, _pu64]),
    "chgpu_filter_description_nullable": (_i, [_vp, _vp, _vp, _pp]),
    "chgpu_sum_add_many": (_i, [_vp, _vp, _u64, _u64, _vp]),
    "chgpu_sum_add_many_conditional": (_i, [_vp, _vp, _vp, _u64, _u64, _vp]),
    "chgpu_filter_sum": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _pu64]),
    "chgpu_filter_sum_async": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "chgpu_and": (_i, [_vp, _vp, _vp, _pp]),
    "chgpu_arith": (_i, [_vp, _i, _vp, _vp, _pp]),
    "chgpu_expr_filter_sum": (_i, [_vp, _u32, _pp, _u32, C.POINTER(_u32), C.POINTER(_i), C.POINTER(_i), _pu64, _i, _u32, _u32, C.POINTER(_i), _vp, _pu64]),
    "chgpu_expr_compile": (_i, [_u32, _vp, _pp]),
    "chgpu_expr_node_type": (_i, [_vp, _u32, C.POINTER(_i)]),
    "chgpu_expr_precompile": (_i, [_vp, _u32, C.POINTER(_u32), _i, _i, _pu64]),
    "chgpu_expr_execute": (_i, [_vp, _vp, _u32, _pp, _u32, C.POINTER(_u32), _pp]),
    "chgpu_expr_filter_sum_node": (_i, [_vp, _vp, _u32, _pp, _i, _i, C.POINTER(_i), _vp, _pu64]),
    "chgpu_expr_filter_execute": (_i, [_vp, _vp, _u32, _pp, _u32, _u32, C.POINTER(_u32), _pp, _pu64]),
    "chgpu_expr_filter_minmax_node": (_i, [_vp, _vp, _u32, _pp, _i, _u32, C.POINTER(_i), _vp, _vp, _pu64]),
    "chgpu_expr_free": (_i, [_vp]),
    "chgpu_index": (_i, [_vp, _vp, _vp, _u64, _i, _pp]),
    "chgpu_replicate": (_i, [_vp, _vp, _vp, _pp]),
    "chgpu_replicate_columns": (_i, [_vp, _u32, _vp, _vp, _vp]),
    "chgpu_sort_permutation": (_i, [_vp, _vp, _vp, _i, _i, _pp]),
    "chgpu_sort_permutation_limit": (_i, [_vp, _vp, _i, _i, _u64, _pp]),
    "chgpu_filter_to_indices": (_i, [_vp, _vp, _pp, _pu64]),
    "chgpu_weak_hash32": (_i, [_vp, _vp, _vp]),
    "chgpu_hash_to_selector": (_i, [_vp, _vp, _u32, _pp]),
    "chgpu_scatter": (_i, [_vp, _vp, _vp, _u32, _pp]),
    "chgpu_partition_by_hash": (_i, [_vp, _vp, _u32, _u32, _pp, _pp, _pu64]),
    "chgpu_pack_fixed_keys": (_i, [_vp, _u32, _pp, _pp]),
    "chgpu_unpack_fixed_key": (_i, [_vp, _vp, _u32, _i, _pp]),
    "chgpu_string_dictionary_encode": (_i, [_vp, _vp, _vp, _pp, _pp, _pu64]),
    "chgpu_string_filter": (_i, [_vp, _vp, _vp, _vp, _pp, _pp, _pu64]),
    "chgpu_lc_remap": (_i, [_vp, _vp, _vp, _pp]),
    "chgpu_agg_create": (_i, [_vp, _i, _u32, C.POINTER(_i), C.POINTER(_i), _u64, _pp]),
    "chgpu_agg_add_block": (_i, [_vp, _vp, _pp, _u64, _u64]),
    "chgpu_agg_add_block_filtered": (_i, [_vp, _vp, _pp, _u64, _u64, _vp]),
    "chgpu_agg_merge": (_i, [_vp, _vp]),
    "chgpu_agg_merge_states": (_i, [_vp, _vp, _pp, _u64]),
    "chgpu_agg_size": (_i, [_vp, _pu64]),
    "chgpu_agg_finalize": (_i, [_vp, _pp, _pp, _pu64]),
    "chgpu_agg_export_states": (_i, [_vp, _pp, _pp, _pu64]),
    "chgpu_agg_export_states_two_level": (_i, [_vp, _pp, _pp, _pu64, _pu64]),
    "chgpu_agg_free": (_i, [_vp]),
    "chgpu_join_create": (_i, [_vp, _i, _i, _i, _i, _u64, _pp]),
    "chgpu_join_add_block": (_i, [_vp, _vp, _vp, _vp, C.POINTER(_u32)]),
    "chgpu_join_finish_build": (_i, [_vp]),
    "chgpu_join_total_rows": (_i, [_vp, _pu64, _pu64]),
    "chgpu_join_probe": (_i, [_vp, _vp, _vp, _u64, _pp, _pp, _pp, _pu64, _pu64]),
    "chgpu_join_probe_agg": (_i, [_vp, _vp, _vp, _vp, _pu64, _vp]),
    "chgpu_compressed_walk_frames": (_i, [_vp, _u64, _i, _u32, C.POINTER(_u32), _pu64, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(_u32)]),
    "chgpu_read_compressed_column": (_i, [_vp, _vp, _u64, _i, _i, _pp]),
    "chgpu_city_hash128": (_i, [_vp, _u64, _pu64]),
    "chgpu_native_walk_block": (_i, [_vp, _u64, _u64, _u32, _vp, C.POINTER(_u32), _pu64, C.POINTER(C.c_int32), C.POINTER(_i), _pu64]),
    "chgpu_native_read_strings": (_i, [_vp, _vp, _u64, _u64, _pp, _pp]),
    "chgpu_col_download_many": (_i, [_vp, _u32, _vp, _vp]),
    "chgpu_asof_create": (_i, [_vp, _i, _i, _i, _i, _pp]),
    "chgpu_asof_add_block": (_i, [_vp, _vp, _vp, _vp, _vp, _pu64]),
    "chgpu_asof_total_rows": (_i, [_vp, _pu64]),
    "chgpu_asof_probe": (_i, [_vp, _vp, _vp, _vp, _pp, _pp, _pu64]),
    "chgpu_asof_free": (_i, [_vp]),
    "chgpu_agg_serialize_states": (_i, [_vp, _i, _vp, _vp, _pp, _pp]),
    "chgpu_agg_deserialize_states": (_i, [_vp, _i, _vp, _u32, _pu64, _pu64, _pp, _pp]),
    "chgpu_fixed_string_word": (_i, [_vp, _vp, _u32, _u32, _pp]),
    "chgpu_fixed_string_from_words": (_i, [_vp, _u32, _pp, _u32, _pp]),
    "chgpu_keydict_create": (_i, [_vp, _u32, _u64, _pp]),
    "chgpu_keydict_encode": (_i, [_vp, _u32, _pp, _u64, _u64, _i, _pp]),
    "chgpu_keydict_size": (_i, [_vp, _pu64]),
    "chgpu_keydict_key_column": (_i, [_vp, _vp, _u32, _i, _pp]),
    "chgpu_keydict_selector": (_i, [_vp, _vp, _u32, _pp]),
    "chgpu_keydict_free": (_i, [_vp]),
    "chgpu_comm_unique_id": (_i, [_vp]),
    "chgpu_comm_init": (_i, [_vp, _i, _i, _vp, _pp]),
    "chgpu_comm_destroy": (_i, [_vp]),
    "chgpu_comm_rank": (_i, [_vp]),
    "chgpu_comm_world": (_i, [_vp]),
    "chgpu_comm_stats": (_i, [_vp, _pu64]),
    "chgpu_all_to_all_counts": (_i, [_vp, _pu64, _pu64]),
    "chgpu_all_to_all": (_i, [_vp, _vp, _pu64, _pu64, _pp]),
    "chgpu_all_to_all_multi": (_i, [_vp, _u32, _pp, _pu64, _pu64, _pp]),
    "chgpu_all_reduce_u64": (_i, [_vp, _vp]),
    "chgpu_all_reduce_u64_host": (_i, [_vp, _pu64, _u32]),
    "chgpu_comm_barrier": (_i, [_vp]),
    "chgpu_join_probe_chain": (_i, [_u32, _pp, _pp, _pp, C.POINTER(_i), _u32, _pp, _pp, _pp, _pp, _pp, _pu64]),
    "chgpu_join_probe_chain_columns": (_i, [_u32, _pp, _pp, _pp, C.POINTER(_i), _pp, _u32, _pp, _pp, _pp, _pp, _pp, _pu64]),
    "chgpu_join_flatten_rowids": (_i, [_vp, _vp, _pp]),
    "chgpu_join_non_joined_rows": (_i, [_vp, _pp, _pu64]),
    "chgpu_join_free": (_i, [_vp]),
}


def declared_symbols(header_path: str = HEADER_PATH):
    """Every function name include/chgpu.h declares (used by the CPU test that checks the exports)."""
    with open(header_path) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(chgpu_[a-z0-9_]+)\s*\(", text)))


class ChgpuError(RuntimeError):
    """Mirror of DB::Exception for this path: .code carries the CHGPU_ERR_* value."""

    def __init__(self, code: int, message: str):
        super().__init__(f"chgpu error {code}: {message}")
        self.code = code


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                "There is no CPU fallback for the product path.")
        # One HIP runtime per process: PyTorch bundles its own libamdhip64.so.7; when it is loaded first the dynamic
        # loader satisfies libchgpu.so's NEEDED libamdhip64.so.7 with that copy.  The other order would bring a second
        # runtime into the process (torch then fails with "no ROCm-capable device"), so import torch first when present.
        try:
            import torch  # noqa: F401  (plumbing only: device memory, streams, torch.distributed)
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here == the .so does not export what the header declares
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int):
    if rc != OK:
        raise ChgpuError(rc, lib().chgpu_last_error().decode("utf-8", "replace"))
