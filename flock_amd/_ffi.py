"""ctypes binding of libflockgpu.so (include/flockgpu.h).

The product path has NO CPU fallback: if the HIP library is missing or a symbol is absent the
import fails loudly.  (The CPU oracle under ``oracle/`` is test infrastructure and is never
imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libflockgpu.so")

OK, ERR_INVALID, ERR_HIP, ERR_OOM, ERR_UNSUPPORTED, ERR_CAPACITY, ERR_PLAN, ERR_PEER = range(8)
H2D, D2H, D2D = 1, 2, 3
ABI_VERSION = 1


class FlockGpuError(RuntimeError):
    """Mirror of FlockError::Execution (flock/src/error.rs:66-68) for the GPU engine."""

    def __init__(self, code: int, message: str):
        super().__init__(f"flockgpu status {code}: {message}")
        self.code = code


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double)]


class BidCols(C.Structure):
    _fields_ = [("auction", C.c_void_p), ("bidder", C.c_void_p), ("price", C.c_void_p),
                ("b_date_time", C.c_void_p), ("rows", C.c_int64)]


class AuctionCols(C.Structure):
    _fields_ = [("a_id", C.c_void_p), ("seller", C.c_void_p), ("category", C.c_void_p), ("rows", C.c_int64)]


class Utf8(C.Structure):
    _fields_ = [("offsets", C.c_void_p), ("data", C.c_void_p)]


class PersonCols(C.Structure):
    _fields_ = [("p_id", C.c_void_p), ("name", Utf8), ("city", Utf8), ("state", Utf8), ("rows", C.c_int64)]


class Windows(C.Structure):
    _fields_ = [("pane_row_offsets", C.POINTER(C.c_int64)), ("n_panes", C.c_int32),
                ("win_pane_lo", C.POINTER(C.c_int32)), ("win_pane_hi", C.POINTER(C.c_int32)),
                ("n_windows", C.c_int32)]


class Q2Result(C.Structure):
    _fields_ = [("auction", C.c_void_p), ("price", C.c_void_p), ("win_out_offsets", C.POINTER(C.c_int64)),
                ("rows", C.c_int64)]


class Q3Result(C.Structure):
    _fields_ = [("name", Utf8), ("city", Utf8), ("state", Utf8), ("a_id", C.c_void_p),
                ("auction_row", C.c_void_p), ("person_row", C.c_void_p),
                ("win_out_offsets", C.POINTER(C.c_int64)), ("rows", C.c_int64),
                ("name_bytes", C.c_int64), ("city_bytes", C.c_int64), ("state_bytes", C.c_int64)]


class Q5Result(C.Structure):
    _fields_ = [("auction", C.c_void_p), ("num", C.c_void_p), ("win_out_offsets", C.POINTER(C.c_int64)),
                ("win_max", C.POINTER(C.c_uint64)), ("win_groups", C.POINTER(C.c_uint64)), ("rows", C.c_int64)]


class Q7Result(C.Structure):
    _fields_ = [("auction", C.c_void_p), ("price", C.c_void_p), ("bidder", C.c_void_p), ("b_date_time", C.c_void_p),
                ("win_out_offsets", C.POINTER(C.c_int64)), ("win_max", C.POINTER(C.c_int64)), ("rows", C.c_int64)]


class AuctionTimeCols(C.Structure):
    _fields_ = [("a_id", C.c_void_p), ("category", C.c_void_p), ("a_date_time", C.c_void_p), ("expires", C.c_void_p),
                ("rows", C.c_int64)]


class Q9Result(C.Structure):
    _fields_ = [("auction", C.c_void_p), ("price", C.c_void_p), ("bidder", C.c_void_p), ("b_date_time", C.c_void_p),
                ("win_out_offsets", C.POINTER(C.c_int64)), ("rows", C.c_int64)]


class Q4Result(C.Structure):
    _fields_ = [("category", C.c_void_p), ("avg_final", C.c_void_p), ("win_out_offsets", C.POINTER(C.c_int64)),
                ("rows", C.c_int64)]


class Q13Result(C.Structure):
    _fields_ = [("auction", C.c_void_p), ("bidder", C.c_void_p), ("price", C.c_void_p), ("b_date_time", C.c_void_p),
                ("value", C.c_void_p), ("bid_row", C.c_void_p), ("side_row", C.c_void_p),
                ("win_out_offsets", C.POINTER(C.c_int64)), ("rows", C.c_int64)]


class Q8Result(C.Structure):
    _fields_ = [("p_id", C.c_void_p), ("name", Utf8), ("person_row", C.c_void_p),
                ("win_out_offsets", C.POINTER(C.c_int64)), ("rows", C.c_int64), ("name_bytes", C.c_int64)]


class PartitionResult(C.Structure):
    _fields_ = [("row", C.c_void_p), ("part_win_offsets", C.POINTER(C.c_int64)), ("rows", C.c_int64)]


class YsbEventCols(C.Structure):
    _fields_ = [("ad_id", Utf8), ("event_type", Utf8), ("rows", C.c_int64)]


class YsbCampaignCols(C.Structure):
    _fields_ = [("c_ad_id", Utf8), ("campaign_id", Utf8), ("rows", C.c_int64)]


class YsbResult(C.Structure):
    _fields_ = [("campaign_id", Utf8), ("count", C.c_void_p), ("win_out_offsets", C.POINTER(C.c_int64)), ("rows", C.c_int64),
                ("campaign_bytes", C.c_int64)]


class IpcBuffer(C.Structure):
    _fields_ = [("data", C.c_void_p), ("bytes", C.c_int64)]


class JsonField(C.Structure):
    _fields_ = [("name", C.c_char_p), ("type", C.c_int32)]


class JsonColumn(C.Structure):
    _fields_ = [("values", C.c_void_p), ("utf8", Utf8), ("utf8_bytes", C.c_int64)]


JSON_INT32, JSON_INT64, JSON_UTF8 = 0, 1, 2


class Q5PartialResult(C.Structure):
    _fields_ = [("auction", C.c_void_p), ("count", C.c_void_p), ("pane_out_offsets", C.POINTER(C.c_int64)), ("rows", C.c_int64)]


class Q11Result(C.Structure):
    _fields_ = [("bidder", C.c_void_p), ("bid_count", C.c_void_p), ("start_time", C.c_void_p), ("end_time", C.c_void_p),
                ("epoch_out_offsets", C.POINTER(C.c_int64)), ("rows", C.c_int64), ("sessions_total", C.c_int64)]


class NexmarkStream(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("first_event_id", C.c_uint64), ("eps", C.c_uint64), ("base_time", C.c_uint64)]


# every symbol include/flockgpu.h declares: (name, restype, argtypes)
_vp, _i, _i64, _u64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64
SYMBOLS = {
    "flockgpu_abi_version": (_i, []),
    "flockgpu_ctx_create": (_i, [_i, _vp, C.POINTER(_vp)]),
    "flockgpu_ctx_destroy": (None, [_vp]),
    "flockgpu_last_error": (C.c_char_p, [_vp]),
    "flockgpu_ctx_synchronize": (_i, [_vp]),
    "flockgpu_malloc": (_i, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "flockgpu_free": (_i, [_vp, _vp]),
    "flockgpu_memcpy": (_i, [_vp, _vp, _vp, C.c_size_t, _i]),
    "flockgpu_malloc_guarded": (_i, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "flockgpu_free_guarded": (_i, [_vp, _vp]),
    "flockgpu_profile_enable": (_i, [_vp, _i]),
    "flockgpu_profile_only": (_i, [_vp, C.c_char_p]),
    "flockgpu_profile_reset": (_i, [_vp]),
    "flockgpu_profile_read": (_i, [_vp, C.POINTER(KernelStat), _i, C.POINTER(_i)]),
    "flockgpu_profile_samples": (_i, [_vp, C.c_char_p, C.POINTER(C.c_float), _i, C.POINTER(_i)]),
    "flockgpu_q1_project": (_i, [_vp, C.POINTER(BidCols), C.c_double, _vp]),
    "flockgpu_q2_filter": (_i, [_vp, C.POINTER(BidCols), C.POINTER(Windows), _i64, C.POINTER(Q2Result)]),
    "flockgpu_q3_join": (_i, [_vp, C.POINTER(AuctionCols), C.POINTER(Windows), C.POINTER(PersonCols),
                              C.POINTER(Windows), _i64, C.POINTER(C.c_char_p), _i, C.POINTER(Q3Result)]),
    "flockgpu_q5_hot_items": (_i, [_vp, C.POINTER(BidCols), C.POINTER(Windows), C.POINTER(Q5Result)]),
    "flockgpu_q7_highest_bid": (_i, [_vp, C.POINTER(BidCols), C.POINTER(Windows), C.POINTER(Q7Result)]),
    "flockgpu_q9_winning_bids": (_i, [_vp, C.POINTER(AuctionTimeCols), C.POINTER(Windows), C.POINTER(BidCols),
                                      C.POINTER(Windows), C.POINTER(Q9Result)]),
    "flockgpu_q4_avg_final_by_category": (_i, [_vp, C.POINTER(AuctionTimeCols), C.POINTER(Windows), C.POINTER(BidCols),
                                               C.POINTER(Windows), C.POINTER(Q4Result)]),
    "flockgpu_q13_side_join": (_i, [_vp, C.POINTER(BidCols), C.POINTER(Windows), _vp, _vp, _i64, C.POINTER(Q13Result)]),
    "flockgpu_q8_join": (_i, [_vp, C.POINTER(PersonCols), C.POINTER(Windows), C.POINTER(AuctionCols),
                              C.POINTER(Windows), C.POINTER(Q8Result)]),
    "flockgpu_partition_by_key": (_i, [_vp, _vp, _i64, C.POINTER(Windows), C.c_int32, C.POINTER(PartitionResult)]),
    "flockgpu_take_i32": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "flockgpu_take_i64": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "flockgpu_take_utf8": (_i, [_vp, C.POINTER(Utf8), _vp, _i64, C.c_int32, C.POINTER(Utf8), C.POINTER(_i64)]),
    "flockgpu_inclusive_scan_i32": (_i, [_vp, _vp, _i64]),
    "flockgpu_ipc_pack_body": (_i, [_vp, C.POINTER(IpcBuffer), _i, _vp, _i64, C.POINTER(_i64)]),
    "flockgpu_json_lines_decode": (_i, [_vp, _vp, _i64, C.POINTER(JsonField), _i, C.POINTER(JsonColumn), C.POINTER(_i64)]),
    "flockgpu_q5_partial_counts": (_i, [_vp, C.POINTER(BidCols), C.POINTER(Windows), C.POINTER(Q5PartialResult)]),
    "flockgpu_q5_hot_items_weighted": (_i, [_vp, _vp, _vp, _i64, C.POINTER(Windows), C.POINTER(Q5Result)]),
    "flockgpu_q11_user_sessions": (_i, [_vp, C.POINTER(BidCols), C.POINTER(_i64), _i, _i, _i64, C.POINTER(Q11Result)]),
    "flockgpu_group_rows_by_key": (_i, [_vp, _vp, _i64, C.POINTER(_vp), C.POINTER(_vp)]),
    "flockgpu_ysb_campaign_counts": (_i, [_vp, C.POINTER(YsbEventCols), C.POINTER(Windows), C.POINTER(YsbCampaignCols), C.c_char_p,
                                          C.POINTER(YsbResult)]),
    "flockgpu_ysb_gen_campaigns": (_i, [_vp, _u64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "flockgpu_ysb_gen_events": (_i, [_vp, _u64, _u64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "flockgpu_nexmark_counts": (_i, [C.POINTER(NexmarkStream), _u64, _u64, C.POINTER(_u64), C.POINTER(_u64),
                                     C.POINTER(_u64)]),
    "flockgpu_nexmark_gen_bids": (_i, [_vp, C.POINTER(NexmarkStream), _u64, _u64, _vp, _vp, _vp, _vp]),
    "flockgpu_nexmark_gen_auctions": (_i, [_vp, C.POINTER(NexmarkStream), _u64, _u64, _vp, _vp, _vp]),
    "flockgpu_nexmark_gen_auction_times": (_i, [_vp, C.POINTER(NexmarkStream), _u64, _u64, _vp, _vp]),
    "flockgpu_nexmark_gen_persons": (_i, [_vp, C.POINTER(NexmarkStream), _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    # include/flockgpu_plan.h (ArrowSchema / ArrowArray travel as raw pointers)
    "flockgpu_plan_create": (_i, [_vp, C.c_char_p, C.c_size_t, C.POINTER(_vp)]),
    "flockgpu_plan_destroy": (None, [_vp]),
    "flockgpu_plan_recognise": (_i, [C.c_char_p, C.c_size_t, C.POINTER(_i)]),
    "flockgpu_plan_query": (_i, [_vp]),
    "flockgpu_plan_num_inputs": (_i, [_vp]),
    "flockgpu_plan_input_name": (C.c_char_p, [_vp, _i]),
    "flockgpu_plan_input_matches": (_i, [_vp, _i, _vp]),
    "flockgpu_plan_feed": (_i, [_vp, _i, _vp, C.POINTER(_vp), _i]),
    "flockgpu_plan_feed_shared": (_i, [_vp, _i, _vp, _i]),
    "flockgpu_plan_execute_retain": (_i, [_vp, C.POINTER(C.c_int64)]),
    "flockgpu_plan_feed_from": (_i, [_vp, _i, _vp]),
    "flockgpu_plan_execute": (_i, [_vp, _vp, _vp]),
    "flockgpu_plan_execute_partitioned": (_i, [_vp, _vp, _vp, _i, C.POINTER(_i)]),
    "flockgpu_plan_explain": (_i, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
    "flockgpu_plan_description": (C.c_char_p, [_vp]),
    "flockgpu_plan_is_shuffling": (_i, [_vp]),
    "flockgpu_plan_output_partitions": (_i, [_vp]),
    # include/flockgpu_comm.h
    "flockgpu_comm_unique_id": (_i, [C.c_char_p]),
    "flockgpu_comm_init_rank": (_i, [_vp, C.c_char_p, _i, _i, C.POINTER(_vp)]),
    "flockgpu_comm_init_local": (_i, [_i, C.POINTER(_vp)]),
    "flockgpu_comm_init_ipc": (_i, [_vp, C.c_char_p, _i, _i, C.POINTER(_vp)]),
    "flockgpu_comm_destroy": (None, [_vp]),
    "flockgpu_comm_rank": (_i, [_vp]),
    "flockgpu_comm_size": (_i, [_vp]),
    "flockgpu_comm_transport": (C.c_char_p, [_vp]),
    "flockgpu_comm_barrier": (_i, [_vp, _vp]),
    "flockgpu_comm_inject_failure": (_i, [_vp, _i]),
    "flockgpu_comm_set_timeout": (_i, [_vp, C.c_double]),
    "flockgpu_comm_set_max_piece_bytes": (_i, [_vp, _i64]),
    "flockgpu_comm_phase_enable": (_i, [_vp, _i]),
    "flockgpu_comm_phase_reset": (_i, [_vp]),
    "flockgpu_comm_phase_read": (_i, [_vp, C.POINTER(KernelStat), _i, C.POINTER(_i)]),
    "flockgpu_q5_hot_items_exchange": (_i, [_vp, _vp, C.POINTER(BidCols), C.POINTER(Windows), C.POINTER(Q5Result)]),
    "flockgpu_q3_join_exchange": (_i, [_vp, _vp, C.POINTER(AuctionCols), C.POINTER(Windows), C.POINTER(PersonCols),
                                       C.POINTER(Windows), _i64, C.POINTER(C.c_char_p), _i, C.POINTER(Q3Result)]),
    "flockgpu_q8_join_exchange": (_i, [_vp, _vp, C.POINTER(PersonCols), C.POINTER(Windows), C.POINTER(AuctionCols),
                                       C.POINTER(Windows), C.POINTER(Q8Result)]),
    "flockgpu_host_alloc": (_i, [C.c_size_t, C.POINTER(_vp)]),
    "flockgpu_host_free": (_i, [_vp]),
    "flockgpu_host_register": (_i, [_vp, C.c_size_t]),
    "flockgpu_host_unregister": (_i, [_vp]),
    "flockgpu_plan_reset": (_i, [_vp]),
    "flockgpu_plan_create_ex": (_i, [_vp, C.c_char_p, C.c_size_t, C.c_uint32, C.POINTER(_vp)]),
    "flockgpu_plan_execute_async": (_i, [_vp, _i]),
    "flockgpu_plan_wait": (_i, [_vp, _vp, _vp, _i, C.POINTER(_i)]),
    "flockgpu_plan_ring_open": (_i, [_vp, _i]),
    "flockgpu_plan_ring_close": (_i, [_vp]),
    "flockgpu_plan_ring_state": (_i, [_vp, C.POINTER(_i64), C.POINTER(_i), C.POINTER(_i)]),
    "flockgpu_plan_feed_pane": (_i, [_vp, _i, _i64, _vp, C.POINTER(_vp), _i]),
    "flockgpu_plan_prefetch_pane": (_i, [_vp, _i, _i64, _vp, C.POINTER(_vp), _i]),
    "flockgpu_plan_partition_scheme": (C.c_char_p, []),
    "flockgpu_plan_check_partition_scheme": (_i, [C.c_char_p]),
    # asynchronous twins of the batched-window calls (one call in flight per ctx)
    "flockgpu_q3_join_async": (_i, [_vp, C.POINTER(AuctionCols), C.POINTER(Windows), C.POINTER(PersonCols),
                                    C.POINTER(Windows), _i64, C.POINTER(C.c_char_p), _i, C.POINTER(Q3Result)]),
    "flockgpu_q5_hot_items_async": (_i, [_vp, C.POINTER(BidCols), C.POINTER(Windows), C.POINTER(Q5Result)]),
    "flockgpu_q8_join_async": (_i, [_vp, C.POINTER(PersonCols), C.POINTER(Windows), C.POINTER(AuctionCols),
                                    C.POINTER(Windows), C.POINTER(Q8Result)]),
    "flockgpu_ctx_wait": (_i, [_vp]),
}
PLAN_GENERIC_ONLY = 1

_lib = None


def load() -> C.CDLL:
    """Loads libflockgpu.so and binds every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m flock_amd.build` (hipcc, gfx950). "
            "flock_amd has no CPU fallback.")
    # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64 / librccl under the same SONAMEs as
    # /opt/rocm's, and whichever is mapped first serves both.  torch is this host's allocator and stream provider, so its
    # runtime must be the one (loading /opt/rocm's first left `torch.cuda` without a device on the GPU boxes).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise ImportError(f"libflockgpu.so does not export {name}")
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.flockgpu_abi_version() != ABI_VERSION:
        raise ImportError("libflockgpu.so ABI version mismatch: rebuild the library")
    _lib = lib
    return lib
