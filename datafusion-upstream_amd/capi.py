"""ctypes binding of include/dfgpu.h (libdfgpu.so) -- the same symbols a Rust shim would bind.

There is no CPU fallback here: if the shared library is missing, or a context cannot be created
because no HIP device is visible, the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdfgpu.so")

# dfgpu_type
BOOL, INT8, INT16, INT32, INT64, UINT8, UINT16, UINT32, UINT64 = 1, 2, 3, 4, 5, 6, 7, 8, 9
FLOAT32, FLOAT64, DATE32, DECIMAL128, UTF8, DICTIONARY = 10, 11, 12, 13, 14, 15
# ops
OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_REM = 0, 1, 2, 3, 4
OP_EQ, OP_NEQ, OP_LT, OP_LTEQ, OP_GT, OP_GTEQ, OP_DISTINCT, OP_NOT_DISTINCT = 10, 11, 12, 13, 14, 15, 16, 17
OP_AND, OP_OR = 20, 21
# join types (datafusion/common/src/join_type.rs:30-47)
JOIN_INNER, JOIN_LEFT, JOIN_RIGHT, JOIN_FULL, JOIN_LEFT_SEMI, JOIN_RIGHT_SEMI, JOIN_LEFT_ANTI, JOIN_RIGHT_ANTI = range(8)
AGG_SUM, AGG_AVG, AGG_COUNT, AGG_MIN, AGG_MAX = range(5)

STATUS_NAMES = {0: "Ok", 1: "Execution", 2: "Internal", 3: "ResourcesExhausted", 4: "NotImplemented", 5: "InvalidArgument"}


class DfgpuError(RuntimeError):
    """Mirrors DataFusionError (common/src/error.rs:52-122): .kind is the variant name."""

    def __init__(self, status: int, message: str):
        self.status = status
        self.kind = STATUS_NAMES.get(status, str(status))
        super().__init__(f"{self.kind} error: {message}")


class ArrayDesc(C.Structure):
    pass


ArrayDesc._fields_ = [
    ("type", C.c_int32), ("precision", C.c_int32), ("scale", C.c_int32), ("key_type", C.c_int32),
    ("length", C.c_int64), ("null_count", C.c_int64),
    ("values", C.c_void_p), ("validity", C.c_void_p), ("offsets", C.c_void_p),
    ("values_bytes", C.c_int64), ("dictionary", C.POINTER(ArrayDesc)),
]


class ArrowSchema(C.Structure):
    pass


ArrowSchema._fields_ = [
    ("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64),
    ("n_children", C.c_int64), ("children", C.POINTER(C.POINTER(ArrowSchema))), ("dictionary", C.POINTER(ArrowSchema)),
    ("release", C.c_void_p), ("private_data", C.c_void_p),
]


class ArrowArray(C.Structure):
    pass


ArrowArray._fields_ = [
    ("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64),
    ("n_children", C.c_int64), ("buffers", C.POINTER(C.c_void_p)), ("children", C.POINTER(C.POINTER(ArrowArray))),
    ("dictionary", C.POINTER(ArrowArray)), ("release", C.c_void_p), ("private_data", C.c_void_p),
]

_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); every symbol declared in include/dfgpu.h
NODE_COLUMN, NODE_SCALAR = -1, -2     # dfgpu_expr_node.op of leaves
PROTOTYPES = {
    "dfgpu_ctx_create": (C.c_int32, [C.c_int32, _P, _PP]),
    "dfgpu_ctx_destroy": (None, [_P]),
    "dfgpu_last_error": (C.c_char_p, [_P]),
    "dfgpu_ctx_synchronize": (C.c_int32, [_P]),
    "dfgpu_ctx_set_option": (C.c_int32, [_P, C.c_char_p, C.c_int64]),
    "dfgpu_acc_update_batch_fused": (C.c_int32, [_P, C.POINTER(_P), C.POINTER(C.c_int32), C.c_int32, _P, C.c_int32, C.POINTER(_P), C.c_int32, _P, _P, C.c_int64]),
    "dfgpu_jit_selftest": (C.c_int32, [C.c_char_p, C.c_char_p, C.c_int64]),
    "dfgpu_take_multi": (C.c_int32, [_P, C.POINTER(_P), C.c_int32, _P, C.POINTER(_P)]),
    "dfgpu_span_begin": (C.c_int32, [_P, C.POINTER(C.c_int64)]),
    "dfgpu_span_end": (C.c_int32, [_P, C.c_int64]),
    "dfgpu_span_elapsed_ns": (C.c_int32, [_P, C.c_int64, C.POINTER(C.c_int64)]),

    "dfgpu_agg_preaggregate": (C.c_int32, [_P, C.POINTER(_P), C.c_int32, C.POINTER(C.c_int32), C.POINTER(_P), C.c_int32, _P, C.POINTER(_P), C.POINTER(_P)]),
    "dfgpu_agg_preaggregate_flags": (C.c_int32, [_P, C.POINTER(_P), C.c_int32, C.POINTER(C.c_int32), C.POINTER(_P), C.POINTER(C.c_int32), C.c_int32, _P, C.c_int32, C.POINTER(_P), C.POINTER(_P)]),
    "dfgpu_ctx_get_option": (C.c_int32, [_P, C.c_char_p, C.POINTER(C.c_int64)]),
    "dfgpu_ctx_stream": (_P, [_P]),
    "dfgpu_version": (C.c_char_p, []),
    "dfgpu_ctx_set_row_selection": (C.c_int32, [_P, _P]),
    "dfgpu_mask_count": (C.c_int32, [_P, _P, C.POINTER(C.c_int64)]),
    "dfgpu_profile_enable": (C.c_int32, [_P, C.c_int32]),
    "dfgpu_profile_select": (C.c_int32, [_P, C.c_char_p]),
    "dfgpu_profile_read": (C.c_int32, [_P, C.c_char_p, C.c_int64]),
    "dfgpu_array_import_host": (C.c_int32, [_P, C.POINTER(ArrayDesc), _PP]),
    "dfgpu_array_wrap_device": (C.c_int32, [_P, C.POINTER(ArrayDesc), _PP]),
    "dfgpu_array_wrap_device_owned": (C.c_int32, [_P, C.POINTER(ArrayDesc), _P, _P, _PP]),
    "dfgpu_array_describe": (C.c_int32, [_P, C.POINTER(ArrayDesc)]),
    "dfgpu_array_export_host": (C.c_int32, [_P, _P, _P, _P, _P]),
    "dfgpu_array_import_arrow": (C.c_int32, [_P, C.POINTER(ArrowArray), C.POINTER(ArrowSchema), _PP]),
    "dfgpu_array_export_arrow": (C.c_int32, [_P, _P, C.POINTER(ArrowArray), C.POINTER(ArrowSchema)]),
    "dfgpu_array_retain": (None, [_P]),
    "dfgpu_array_release": (None, [_P]),
    "dfgpu_array_is_identity": (C.c_int32, [_P]),
    "dfgpu_array_length": (C.c_int64, [_P]),
    "dfgpu_array_null_count": (C.c_int64, [_P, _P]),
    "dfgpu_array_slice": (C.c_int32, [_P, _P, C.c_int64, C.c_int64, _PP]),
    "dfgpu_concat": (C.c_int32, [_P, _PP, C.c_int32, _PP]),
    "dfgpu_list_from_counts": (C.c_int32, [_P, _P, _P, _PP]),
    "dfgpu_list_flatten": (C.c_int32, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _PP, _PP]),
    "dfgpu_array_new_null": (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int64, _PP]),
    "dfgpu_array_new_zeros": (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int64, _PP]),
    "dfgpu_array_make_dictionary": (C.c_int32, [_P, _P, _P, _PP]),
    "dfgpu_hash_columns": (C.c_int32, [_P, _PP, C.c_int32, C.c_uint64, _PP]),
    "dfgpu_take": (C.c_int32, [_P, _P, _P, _PP]),
    "dfgpu_filter": (C.c_int32, [_P, _P, _P, _PP]),
    "dfgpu_mask_to_indices": (C.c_int32, [_P, _P, _PP]),
    "dfgpu_binary": (C.c_int32, [_P, C.c_int32, _P, C.c_int32, _P, C.c_int32, _PP]),
    "dfgpu_binary_fused2": (C.c_int32, [_P, C.c_int32, _P, C.c_int32, _P, _P, C.c_int32, C.c_int32, _PP]),
    "dfgpu_not": (C.c_int32, [_P, _P, _PP]),
    "dfgpu_is_null": (C.c_int32, [_P, _P, C.c_int32, _PP]),
    "dfgpu_negative": (C.c_int32, [_P, _P, _PP]),
    "dfgpu_cast": (C.c_int32, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _PP]),
    "dfgpu_in_list": (C.c_int32, [_P, _P, _P, C.c_int32, _PP]),
    "dfgpu_join_build": (C.c_int32, [_P, _PP, C.c_int32, _P, C.c_int32, _PP]),
    "dfgpu_join_table_free": (None, [_P]),
    "dfgpu_join_table_num_rows": (C.c_int64, [_P]),
    "dfgpu_join_table_memory": (C.c_int64, [_P]),
    "dfgpu_join_probe": (C.c_int32, [_P, _P, _PP, C.c_int32, _P, _PP, _PP]),
    "dfgpu_join_probe_deferred": (C.c_int32, [_P, _P, _PP, C.c_int32, _P, _PP, _PP]),
    "dfgpu_join_lookup": (C.c_int32, [_P, _P, _PP, C.c_int32, _P, _PP]),
    "dfgpu_join_probe_selection": (C.c_int32, [_P, _P, _PP, C.c_int32, _P, _PP]),
    "dfgpu_join_mark_visited": (C.c_int32, [_P, _P, _P]),
    "dfgpu_join_adjust_indices": (C.c_int32, [_P, _P, _P, C.c_int64, C.c_int64, C.c_int32, _PP, _PP]),
    "dfgpu_join_final_indices": (C.c_int32, [_P, _P, C.c_int32, _PP]),
    "dfgpu_groups_new": (C.c_int32, [_P, C.c_int32, _PP]),
    "dfgpu_groups_free": (None, [_P]),
    "dfgpu_groups_intern": (C.c_int32, [_P, _P, _PP, C.c_int32, _P, _PP]),
    "dfgpu_groups_intern_deferred": (C.c_int32, [_P, _P, _PP, C.c_int32, _P, _PP]),
    "dfgpu_groups_len": (C.c_int64, [_P]),
    "dfgpu_groups_size": (C.c_int64, [_P]),
    "dfgpu_groups_emit": (C.c_int32, [_P, _P, _PP]),
    "dfgpu_groups_emit_deferred": (C.c_int32, [_P, _P, _PP, _PP]),
    "dfgpu_acc_new": (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _PP]),
    "dfgpu_acc_free": (None, [_P]),
    "dfgpu_acc_update_batch": (C.c_int32, [_P, _P, _P, _P, _P, C.c_int64]),
    "dfgpu_acc_update_batch_multi": (C.c_int32, [_P, _PP, _PP, _PP, C.c_int32, _P, C.c_int64]),
    "dfgpu_acc_merge_batch": (C.c_int32, [_P, _P, _PP, C.c_int32, _P, _P, C.c_int64]),
    "dfgpu_acc_evaluate": (C.c_int32, [_P, _P, _PP]),
    "dfgpu_acc_state": (C.c_int32, [_P, _P, _PP, C.POINTER(C.c_int32)]),
    "dfgpu_acc_size": (C.c_int64, [_P]),
    "dfgpu_acc_emit_first": (C.c_int32, [_P, _P, C.c_int64, C.c_int32, _PP, C.POINTER(C.c_int32)]),
    "dfgpu_groups_emit_first": (C.c_int32, [_P, _P, C.c_int64, _PP]),
    "dfgpu_array_iota": (C.c_int32, [_P, C.c_int64, _PP]),
    "dfgpu_cross_join_indices": (C.c_int32, [_P, C.c_int64, C.c_int64, C.c_int64, C.c_int32, _PP, _PP]),
    "dfgpu_sort_to_indices": (C.c_int32, [_P, _PP, C.c_char_p, C.c_char_p, C.c_int32, C.c_int64, _PP]),
    "dfgpu_sort_to_indices_keys": (C.c_int32, [_P, _PP, C.c_char_p, C.c_char_p, C.c_int32, C.c_int64, _PP, _PP]),
    "dfgpu_sort_take": (C.c_int32, [_P, _PP, C.c_char_p, C.c_char_p, C.c_int32, C.c_int64, _PP, C.c_int32, _PP, _PP, _PP]),
    "dfgpu_hash_partition": (C.c_int32, [_P, _PP, C.c_int32, C.c_int32, _PP, C.POINTER(C.c_int64)]),
    "dfgpu_comm_unique_id": (C.c_int32, [C.c_char_p]),
    "dfgpu_comm_create_rccl": (C.c_int32, [_P, C.c_char_p, C.c_int32, C.c_int32, _PP]),
    "dfgpu_comm_create_custom": (C.c_int32, [_P, _PP]),
    "dfgpu_comm_free": (None, [_P]),
    "dfgpu_comm_rank": (C.c_int32, [_P]),
    "dfgpu_comm_world": (C.c_int32, [_P]),
    "dfgpu_exchange": (C.c_int32, [_P, _P, _PP, C.c_int32, _PP, C.c_int32, _P, _PP, C.POINTER(C.c_int64)]),
    "dfgpu_partition_columns": (C.c_int32, [_P, _PP, C.c_int32, C.c_int32, _PP, C.c_int32, _P, _PP, _PP, C.POINTER(C.c_int64)]),
    "dfgpu_csv_read": (C.c_int32, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, _PP, C.POINTER(C.c_int64)]),
    "dfgpu_parquet_open": (C.c_int32, [_P, _P, C.c_int64, _P, _PP]),
    "dfgpu_parquet_open_file": (C.c_int32, [_P, C.c_char_p, C.c_int32, _PP]),
    "dfgpu_parquet_close": (None, [_P]),
    "dfgpu_parquet_set_option": (C.c_int32, [_P, C.c_char_p, C.c_int64]),
    "dfgpu_parquet_num_rows": (C.c_int64, [_P]),
    "dfgpu_parquet_num_row_groups": (C.c_int32, [_P]),
    "dfgpu_parquet_num_columns": (C.c_int32, [_P]),
    "dfgpu_parquet_row_group_rows": (C.c_int64, [_P, C.c_int32]),
    "dfgpu_parquet_column_name": (C.c_char_p, [_P, C.c_int32]),
    "dfgpu_parquet_column_type": (C.c_int32, [_P, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "dfgpu_parquet_column_stats": (C.c_int32, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "dfgpu_parquet_column_chunk_bytes": (C.c_int64, [_P, C.c_int32, C.c_int32, C.c_int32]),
    "dfgpu_parquet_read": (C.c_int32, [_P, _P, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32, _PP]),
}

# include/dfgpu_exec.h: the C++ host layer (ExecutionPlan / PhysicalExpr mirror)
_CPP = C.POINTER(C.c_char_p)
_I32P = C.POINTER(C.c_int32)
PROTOTYPES.update({
    "dfgpu_exec_last_error": (C.c_char_p, []),
    "dfgpu_plan_sort_merge_join": (C.c_int32, [_P, _P, _PP, _PP, C.c_int32, _P, _I32P, _I32P, C.c_int32, C.c_int32, C.c_int32, _PP]),
    "dfgpu_plan_nested_loop_join": (C.c_int32, [_P, _P, _P, _I32P, _I32P, C.c_int32, C.c_int32, _PP]),
    "dfgpu_plan_parquet": (C.c_int32, [_P, _I32P, C.c_int32, C.c_int32, C.c_int32, _PP]),
    "dfgpu_plan_csv": (C.c_int32, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_char_p), _I32P, C.c_int32, _I32P, C.c_int32, C.c_int32, C.c_int64, _PP]),
    "dfgpu_plan_parquet_prune": (C.c_int32, [_P, C.c_int32, C.c_int64, C.c_int64]),
    "dfgpu_plan_parquet_pruned": (C.c_int64, [_P]),
    "dfgpu_batch_new": (C.c_int32, [_CPP, _PP, C.c_int32, _PP]),
    "dfgpu_batch_free": (None, [_P]),
    "dfgpu_batch_num_columns": (C.c_int32, [_P]),
    "dfgpu_batch_num_rows": (C.c_int32, [_P, _P, C.POINTER(C.c_int64)]),
    "dfgpu_batch_column_name": (C.c_char_p, [_P, C.c_int32]),
    "dfgpu_batch_materialize": (C.c_int32, [_P, _P]),
    "dfgpu_batch_column": (C.c_int32, [_P, _P, C.c_int32, _PP]),
    "dfgpu_expr_column": (C.c_int32, [C.c_char_p, C.c_int32, _PP]),
    "dfgpu_expr_literal": (C.c_int32, [_P, _PP]),
    "dfgpu_expr_binary": (C.c_int32, [_P, C.c_int32, _P, _PP]),
    "dfgpu_expr_not": (C.c_int32, [_P, _PP]),
    "dfgpu_expr_is_null": (C.c_int32, [_P, C.c_int32, _PP]),
    "dfgpu_expr_negative": (C.c_int32, [_P, _PP]),
    "dfgpu_expr_cast": (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_int32, _PP]),
    "dfgpu_expr_in_list": (C.c_int32, [_P, _P, C.c_int32, _PP]),
    "dfgpu_expr_free": (None, [_P]),
    "dfgpu_plan_memory": (C.c_int32, [_PP, _I32P, C.c_int32, _PP]),
    "dfgpu_plan_memory_replace": (C.c_int32, [_P, _PP, _I32P, C.c_int32]),
    "dfgpu_plan_filter": (C.c_int32, [_P, _P, _PP]),
    "dfgpu_plan_projection": (C.c_int32, [_PP, _CPP, C.c_int32, _P, _PP]),
    "dfgpu_plan_coalesce_batches": (C.c_int32, [_P, C.c_int64, _PP]),
    "dfgpu_plan_coalesce_partitions": (C.c_int32, [_P, _PP]),
    "dfgpu_plan_repartition": (C.c_int32, [_P, _PP, C.c_int32, C.c_int32, _PP]),
    "dfgpu_plan_hash_join": (C.c_int32, [_P, _P, _PP, _PP, C.c_int32, _P, _I32P, _I32P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _PP]),
    "dfgpu_plan_aggregate_input_order": (C.c_int32, [_P, C.c_int32, _I32P, C.c_int32]),
    "dfgpu_plan_aggregate_grouping_sets": (C.c_int32, [_P, _PP, C.c_int32, C.POINTER(C.c_uint8), C.c_int32]),
    "dfgpu_plan_aggregate": (C.c_int32, [C.c_int32, _PP, _CPP, C.c_int32, _I32P, _PP, _PP, _CPP, _I32P, C.c_int32, _P, _PP]),
    "dfgpu_plan_sort": (C.c_int32, [_PP, C.c_char_p, C.c_char_p, C.c_int32, C.c_int64, C.c_int32, _P, _PP]),
    "dfgpu_plan_sort_preserving_merge": (C.c_int32, [_PP, C.c_char_p, C.c_char_p, C.c_int32, C.c_int64, _P, _PP]),
    "dfgpu_plan_free": (None, [_P]),
    "dfgpu_plan_with_fresh_state": (C.c_int32, [_P, _PP]),
    "dfgpu_plan_partition_count": (C.c_int32, [_P]),
    "dfgpu_plan_schema_len": (C.c_int32, [_P]),
    "dfgpu_plan_schema_name": (C.c_char_p, [_P, C.c_int32]),
    "dfgpu_plan_name": (C.c_char_p, [_P]),
    "dfgpu_plan_metrics": (C.c_int32, [_P, C.c_char_p, C.c_int64]),
    "dfgpu_plan_execute": (C.c_int32, [_P, C.c_int32, _P, C.c_int64, _PP]),
    "dfgpu_stream_next": (C.c_int32, [_P, _PP]),
    "dfgpu_stream_free": (None, [_P]),
})

_lib = None


def load_library() -> C.CDLL:
    """Load libdfgpu.so (built in-tree by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'); "
                          "there is no CPU fallback for the dfgpu operators")
    # torch bundles its own ROCm runtime (libamdhip64): when torch is used in the same process (device memory,
    # streams, RCCL plumbing) it must be loaded first so both sides share ONE HIP runtime; loading the system
    # runtime first makes torch report "No HIP GPUs are available".
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)      # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def handle_array(ptrs):
    """list of raw handles -> (ctypes array of void*, n)"""
    arr = (C.c_void_p * max(len(ptrs), 1))(*ptrs)
    return arr, len(ptrs)
