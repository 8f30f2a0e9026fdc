"""ctypes binding of libtgpu.so (include/tgpu.h).  No CPU fallback: a missing or stale library is a hard error."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libtgpu.so")
CSRC = os.path.join(_HERE, "csrc")


class TgpuError(RuntimeError):
    """Mirror of io.trino.spi.TrinoException: `code` is the tgpu.h status (StandardErrorCode mirror)."""

    NAMES = {-1: "INVALID_ARGUMENT", -2: "NUMERIC_VALUE_OUT_OF_RANGE", -3: "GENERIC_INSUFFICIENT_RESOURCES", -4: "COMPILER_ERROR",
             -5: "GENERIC_INTERNAL_ERROR", -6: "DEVICE_ERROR", -7: "DIVISION_BY_ZERO", -8: "NOT_SUPPORTED", -9: "INVALID_CAST_ARGUMENT"}

    def __init__(self, code, message):
        super().__init__(f"{self.NAMES.get(code, code)}: {message}")
        self.code = code
        self.message = message


def build(force=False, jobs=8):
    """Compile every HIP source of the package for gfx950 into libtgpu.so (in-tree)."""
    args = ["make", "-C", CSRC, f"-j{jobs}"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return SO_PATH


class Block(C.Structure):
    pass


Block._fields_ = [("type", C.c_int32), ("encoding", C.c_int32), ("memory", C.c_int32), ("position_count", C.c_int32),
                  ("values", C.c_void_p), ("nulls", C.c_void_p), ("offsets", C.c_void_p), ("ids", C.c_void_p),
                  ("dictionary", C.POINTER(Block))]


class Page(C.Structure):
    _fields_ = [("position_count", C.c_int32), ("channel_count", C.c_int32), ("blocks", C.POINTER(Block))]


class ExprNode(C.Structure):
    _fields_ = [("kind", C.c_int32), ("type", C.c_int32), ("op", C.c_int32), ("n_args", C.c_int32), ("args", C.c_int32 * 3),
                ("is_null", C.c_int32), ("ival", C.c_int64), ("dval", C.c_double), ("slen", C.c_int32), ("pad", C.c_int32)]


class PageProcessorSpec(C.Structure):
    _fields_ = [("nodes", C.POINTER(ExprNode)), ("node_count", C.c_int32), ("string_pool", C.c_char_p), ("string_pool_len", C.c_int32),
                ("filter_root", C.c_int32), ("projection_count", C.c_int32), ("projection_roots", C.POINTER(C.c_int32))]


TRANSPORT_META_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32)
TRANSPORT_V_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_int64))


class ExchangeTransport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_to_all_meta", TRANSPORT_META_FN), ("all_to_all_v", TRANSPORT_V_FN)]


SOURCE_NEXT_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(Page))
SOURCE_FLAG_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p)
SOURCE_LOAD_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.POINTER(Block))
SOURCE_CLOSE_FN = C.CFUNCTYPE(None, C.c_void_p)


class PageSource(C.Structure):
    _fields_ = [("user", C.c_void_p), ("get_next_page", SOURCE_NEXT_FN), ("is_finished", SOURCE_FLAG_FN), ("is_blocked", SOURCE_FLAG_FN), ("load_block", SOURCE_LOAD_FN),
                ("close", SOURCE_CLOSE_FN)]


CURSOR_FIELD_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32)
CURSOR_LONG_FN = C.CFUNCTYPE(C.c_int64, C.c_void_p, C.c_int32)
CURSOR_DOUBLE_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_int32)
CURSOR_SLICE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32))
CURSOR_BYTES_FN = C.CFUNCTYPE(C.c_int64, C.c_void_p)


class RecordCursor(C.Structure):
    _fields_ = [("user", C.c_void_p), ("advance_next_position", SOURCE_FLAG_FN), ("is_null", CURSOR_FIELD_FN), ("get_boolean", CURSOR_FIELD_FN), ("get_long", CURSOR_LONG_FN),
                ("get_double", CURSOR_DOUBLE_FN), ("get_slice", CURSOR_SLICE_FN), ("completed_bytes", CURSOR_BYTES_FN), ("close", SOURCE_CLOSE_FN)]


class AggSpec(C.Structure):
    _fields_ = [("function", C.c_int32), ("input_channel", C.c_int32), ("mask_channel", C.c_int32)]


_lib = None

# every symbol include/tgpu.h declares: (restype, argtypes)
i32, i64, vp, cp = C.c_int32, C.c_int64, C.c_void_p, C.c_char_p
P = C.POINTER
SYMBOLS = {
    "tgpu_context_create": (i32, [i32, vp, P(vp)]),
    "tgpu_context_destroy": (None, [vp]),
    "tgpu_context_synchronize": (i32, [vp]),
    "tgpu_last_error": (cp, []),
    "tgpu_version": (cp, []),
    "tgpu_set_resource_dir": (i32, [cp]),
    "tgpu_context_set_double_sum_order": (i32, [vp, i32]),
    "tgpu_context_set_device_input_stable": (i32, [vp, i32]),
    "tgpu_orc_decode_long_column": (i32, [vp, i32, i32, i32, vp, i64, vp, i64, vp]),
    "tgpu_orc_decode_boolean_column": (i32, [vp, i32, vp, i64, vp, i64, vp]),
    "tgpu_orc_decode_dictionary_string_column": (i32, [vp, i32, i32, vp, i64, vp, i64, i32, vp, i64, vp, i64, vp]),
    "tgpu_orc_decode_direct_string_column": (i32, [vp, i32, i32, vp, i64, vp, i64, vp, i64, vp]),
    "tgpu_orc_decode_double_column": (i32, [vp, i32, vp, i64, vp, i64, vp]),
    "tgpu_parquet_decode_data_page": (i32, [vp, i32, i32, i32, i32, vp, i64, vp, i64, vp, i64, i32, vp]),
    "tgpu_context_set_max_output_page": (i32, [vp, i64, i64]),
    "tgpu_pinned_alloc": (i32, [vp, i64, P(vp)]),
    "tgpu_pinned_free": (i32, [vp, vp]),
    "tgpu_profile_enable": (i32, [vp, i32]),
    "tgpu_profile_reset": (i32, [vp]),
    "tgpu_profile_dump": (i64, [vp, cp, i64]),
    "tgpu_filter_project_factory_create": (i32, [vp, i32, i32, P(i32), P(PageProcessorSpec), P(vp)]),
    "tgpu_hash_aggregation_factory_create": (i32, [vp, i32, i32, P(i32), P(i32), i32, i32, i32, P(AggSpec), i32, i32, P(vp)]),
    "tgpu_hash_builder_factory_create": (i32, [vp, i32, i32, P(i32), i32, P(i32), i32, P(i32), i32, i32, P(vp), P(vp)]),
    "tgpu_partitioned_hash_builder_factory_create": (i32, [vp, i32, i32, P(i32), i32, P(i32), i32, P(i32), i32, i32, i32, P(vp), P(vp)]),
    "tgpu_partitioned_join_position_encode": (i64, [i32, i32, i32]),
    "tgpu_partitioned_join_position_decode": (i32, [i64, i32, P(i32), P(i32)]),
    "tgpu_lookup_source_factory_destroy": (None, [vp]),
    "tgpu_lookup_source_stats": (i32, [vp, P(i64), P(i64), P(i64)]),
    "tgpu_lookup_source_factory_set_join_filter": (i32, [vp, i32, P(i32), P(PageProcessorSpec)]),
    "tgpu_lookup_join_factory_create": (i32, [vp, i32, vp, i32, P(i32), i32, P(i32), i32, i32, P(i32), i32, P(vp)]),
    "tgpu_top_n_factory_create": (i32, [vp, i32, i32, P(i32), C.c_int64, i32, P(i32), P(i32), P(vp)]),
    "tgpu_order_by_factory_create": (i32, [vp, i32, i32, P(i32), i32, P(i32), i32, i32, P(i32), P(i32), P(vp)]),
    "tgpu_filter_project_lookup_join_factory_create": (i32, [vp, i32, vp, i32, P(i32), P(PageProcessorSpec), i32, P(i32), i32, i32, P(i32), i32, P(vp)]),
    "tgpu_filter_project_hash_aggregation_factory_create": (i32, [vp, i32, i32, P(i32), P(PageProcessorSpec), i32, P(i32), P(i32), i32, i32, i32, P(AggSpec), i32, P(vp)]),
    "tgpu_scan_filter_project_factory_create": (i32, [vp, i32, i32, P(i32), P(PageProcessorSpec), P(vp)]),
    "tgpu_scan_operator_add_page_source": (i32, [vp, P(PageSource)]),
    "tgpu_scan_operator_add_record_cursor": (i32, [vp, P(RecordCursor), i32, P(i32)]),
    "tgpu_scan_operator_no_more_splits": (i32, [vp]),
    "tgpu_scan_operator_stats": (i32, [vp, P(i64), P(i64), P(i64)]),
    "tgpu_operator_factory_create_operator": (i32, [vp, P(vp)]),
    "tgpu_operator_factory_no_more_operators": (i32, [vp]),
    "tgpu_operator_factory_duplicate": (i32, [vp, P(vp)]),
    "tgpu_operator_start_memory_revoke": (i32, [vp]),
    "tgpu_operator_finish_memory_revoke": (i32, [vp]),
    "tgpu_operator_revocable_memory_bytes": (i64, [vp]),
    "tgpu_operator_spill_stats": (i32, [vp, P(i64), P(i64)]),
    "tgpu_hash_aggregation_factory_set_spill_enabled": (i32, [vp, i32]),
    "tgpu_hash_aggregation_factory_set_max_partial_memory": (i32, [vp, i64]),
    "tgpu_operator_factory_destroy": (None, [vp]),
    "tgpu_operator_needs_input": (i32, [vp]),
    "tgpu_operator_add_input": (i32, [vp, P(Page)]),
    "tgpu_operator_get_output": (i32, [vp, P(vp)]),
    "tgpu_operator_finish": (i32, [vp]),
    "tgpu_operator_is_finished": (i32, [vp]),
    "tgpu_operator_is_blocked": (i32, [vp]),
    "tgpu_operator_memory_bytes": (i64, [vp]),
    "tgpu_operator_close": (None, [vp]),
    "tgpu_output_page_position_count": (i32, [vp]),
    "tgpu_output_page_channel_count": (i32, [vp]),
    "tgpu_output_page_as_page": (i32, [vp, P(Page)]),
    "tgpu_output_page_block_info": (i32, [vp, i32, P(i32), P(i64), P(i32)]),
    "tgpu_output_page_copy_block": (i32, [vp, i32, vp, vp, vp]),
    "tgpu_output_page_copy_blocks": (i32, [vp, i32, vp, vp, vp]),
    "tgpu_output_page_release": (None, [vp]),
    "tgpu_group_by_hash_create": (i32, [vp, i32, P(i32), P(i32), i32, i32, P(vp)]),
    "tgpu_group_by_hash_destroy": (None, [vp]),
    "tgpu_group_by_hash_add_page": (i32, [vp, P(Page)]),
    "tgpu_group_by_hash_get_group_ids": (i32, [vp, P(Page), vp, P(i64)]),
    "tgpu_group_by_hash_contains": (i32, [vp, i32, P(Page), P(i32)]),
    "tgpu_group_by_hash_group_count": (i64, [vp]),
    "tgpu_group_by_hash_capacity": (i32, [vp]),
    "tgpu_group_by_hash_estimated_size": (i64, [vp]),
    "tgpu_group_by_hash_append_values": (i32, [vp, P(vp)]),
    "tgpu_hash_page": (i32, [vp, P(Page), i32, P(i32), vp]),
    "tgpu_partition_page": (i32, [vp, P(Page), i32, P(i32), i32, i32, vp, P(vp)]),
    "tgpu_operator_add_input_output_page": (i32, [vp, vp]),
    "tgpu_lookup_outer_factory_create": (i32, [vp, i32, vp, i32, P(i32), P(vp)]),
    "tgpu_dynamic_filter_source_factory_create": (i32, [vp, i32, i32, P(i32), i32, P(i32), i32, i64, i32, P(vp)]),
    "tgpu_dynamic_filter_source_result": (i32, [vp, i32, P(i32), P(vp), P(i64), P(i64)]),
    "tgpu_merge_pages_factory_create": (i32, [vp, i32, i32, P(i32), i64, i32, i64, P(vp)]),
    "tgpu_partitioned_output_factory_create": (i32, [vp, i32, i32, P(i32), i32, P(i32), i32, i32, i32, i32, i32, P(vp)]),
    "tgpu_partitioned_output_poll": (i32, [vp, P(i32), P(vp)]),
    "tgpu_partitioned_output_info": (i32, [vp, P(i64), P(i64)]),
    "tgpu_serialize_page": (i32, [vp, P(Page), vp, i64, P(i64)]),
    "tgpu_deserialize_page": (i32, [vp, vp, i64, i32, P(i32), P(vp)]),
    "tgpu_exchange_unique_id": (i32, [vp]),
    "tgpu_exchange_create": (i32, [vp, cp, i32, i32, P(vp)]),
    "tgpu_exchange_create_with_transport": (i32, [vp, i32, i32, P(ExchangeTransport), P(vp)]),
    "tgpu_exchange_destroy": (None, [vp]),
    "tgpu_exchange_repartition": (i32, [vp, P(Page), i32, P(i32), i32, P(vp)]),
    "tgpu_exchange_partitioned_output": (i32, [vp, vp, i32, P(i32), P(vp)]),
    "tgpu_exchange_all_gather": (i32, [vp, P(Page), P(vp)]),
    "tgpu_exchange_bytes_sent": (i64, [vp]),
}
# helpers outside tgpu.h (build / diagnostics)
EXTRA_SYMBOLS = {
    "tgpu_precompile_page_processor": (i32, [i32, P(i32), P(PageProcessorSpec)]),
    "tgpu_page_processor_source": (i64, [i32, P(i32), P(PageProcessorSpec), cp, i64]),
    "tgpu_group_by_hash_rehash_count": (i32, [vp]),
    "tgpu_debug_bind_count": (C.c_longlong, []),
    "tgpu_debug_dictionary_pages": (i64, [vp]),
    "tgpu_precompile_fused_probe": (i32, [i32, P(i32), P(PageProcessorSpec), i32, i32, P(i32)]),
    "tgpu_precompile_fused_aggregation": (i32, [i32, P(i32), P(PageProcessorSpec), i32, P(AggSpec), i32, P(i32)]),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise TgpuError(-6, f"{SO_PATH} is missing: build it with `python -c \"import __graft_entry__ as g; g.build()\"` "
                                "(there is no CPU fallback)")
        L = C.CDLL(SO_PATH)
        for table in (SYMBOLS, EXTRA_SYMBOLS):
            for name, (res, args) in table.items():
                f = getattr(L, name)
                f.restype = res
                f.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc < 0:
        raise TgpuError(rc, lib().tgpu_last_error().decode("utf-8", "replace"))
    return rc
