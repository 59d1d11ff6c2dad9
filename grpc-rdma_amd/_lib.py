"""ctypes binding of libgrdma_amd.so (the C ABI in include/grdma_amd.h).

The library is the product; this module only loads it.  There is no Python or
CPU fallback: if the shared object is missing the import of any symbol raises,
and on a machine without a HIP device `init()` raises `GrdmaError`.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GRDMA_LIB_PATH") or os.path.join(HERE, "libgrdma_amd.so")  # (override: A/B builds in tools/)

u64 = C.c_uint64
u64p = C.POINTER(C.c_uint64)


class GrdmaError(RuntimeError):
    pass


class Slice(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("len", u64)]


class ReadSlice(C.Structure):
    _fields_ = [("off", u64), ("len", u64)]


class PairState(C.Structure):
    _fields_ = [(n, u64) for n in (
        "head", "moving_head", "remain", "remote_tail", "remote_head", "internal_read_size",
        "credit_msgs", "partial_write", "total_read", "total_written", "leftover_cap")]


class Config(C.Structure):
    _fields_ = [("device_name", C.c_char * 64), ("port_num", C.c_int32), ("gid_index", C.c_int32),
                ("poller_thread_num", C.c_int32), ("busy_polling_timeout_us", C.c_int32),
                ("poller_sleep_timeout_ms", C.c_int32), ("ring_buffer_size_kb", C.c_uint32),
                ("zerocopy_buffer_size_kb", C.c_uint32), ("zerocopy_threshold_kb", C.c_uint32),
                ("max_sge", C.c_int32), ("hip_device", C.c_int32), ("hip_wire_direct", C.c_int32),
                ("hip_register_min", C.c_uint32), ("hip_pair_pool_mb", C.c_uint32)]


# name -> (restype, argtypes); kept in one table so the "exports every symbol"
# test can walk it against include/grdma_amd.h.
SIGNATURES = {
    "grdma_abi_version": (C.c_int, []),
    "grdma_stream_job_set_rebuild_index": (C.c_int, [C.c_void_p, C.c_int]),
    "grdma_verbs_supported": (C.c_int, []),
    "grdma_pair_verbs_open": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int]),
    "grdma_pair_verbs_address": (C.c_int, [C.c_void_p, C.c_void_p]),
    "grdma_pair_verbs_connect": (C.c_int, [C.c_void_p, C.c_void_p]),
    "grdma_pair_verbs_counts": (C.c_int, [C.c_void_p, C.POINTER(u64)]),
    "grdma_pair_watch_hits": (C.c_int64, [C.c_void_p]),
    "grdma_pingpong_end": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Slice), u64, u64, C.c_int, u64, u64, C.POINTER(u64), C.POINTER(u64)]),
    "grdma_engine_watchers": (C.c_int, []),
    "grdma_watch_fast_drains": (u64, []),
    "grdma_watch_ticks": (C.c_int, [C.POINTER(u64)]),
    "grdma_parse_platform": (C.c_int, [C.c_char_p]),
    "grdma_determine_platform": (C.c_int, []),
    "grdma_config_from_env": (C.c_int, [C.POINTER(Config)]),
    "grdma_init": (C.c_int, [C.c_int]),
    "grdma_poller_create": (C.c_void_p, [C.c_int, C.c_int]),
    "grdma_poller_destroy": (None, [C.c_void_p]),
    "grdma_poller_add": (C.c_int, [C.c_void_p, C.c_void_p]),
    "grdma_poller_remove": (C.c_int, [C.c_void_p, C.c_void_p]),
    "grdma_poller_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "grdma_pair_get_wakeup_fd": (C.c_int, [C.c_void_p]),
    "grdma_pair_consume_wakeup": (C.c_int, [C.c_void_p]),
    "grdma_device_count": (C.c_int, []),
    "grdma_last_error": (C.c_char_p, []),
    "grdma_pair_create": (C.c_void_p, [u64, C.c_int, C.c_int]),
    "grdma_pair_connect": (C.c_int, [C.c_void_p, C.c_void_p]),
    "grdma_pair_disconnect": (C.c_int, [C.c_void_p]),
    "grdma_pair_export_address": (C.c_int, [C.c_void_p, C.c_void_p]),
    "grdma_pair_connect_remote": (C.c_int, [C.c_void_p, C.c_void_p]),
    "grdma_pair_bootstrap_fd": (C.c_int, [C.c_void_p, C.c_int]),
    "grdma_pair_destroy": (None, [C.c_void_p]),
    "grdma_pair_get_status": (C.c_int, [C.c_void_p]),
    "grdma_pair_send": (C.c_int64, [C.c_void_p, C.POINTER(Slice), u64, u64, C.c_int]),
    "grdma_pair_recv": (C.c_int64, [C.c_void_p, C.c_void_p, u64, C.c_int]),
    "grdma_pair_enable_zerocopy": (C.c_int, [C.c_void_p, u64]),
    "grdma_pair_enable_zerocopy_ex": (C.c_int, [C.c_void_p, u64, C.c_int]),
    "grdma_pair_zerocopy_mem": (C.c_int, [C.c_void_p]),
    "grdma_pair_allocate_send_buffer": (C.c_void_p, [C.c_void_p, u64]),
    "grdma_pair_send_zerocopy": (C.c_int64, [C.c_void_p, C.POINTER(Slice), u64, u64, C.c_int]),
    "grdma_pair_zerocopy_state": (C.c_int, [C.c_void_p, u64p]),
    "grdma_pair_has_message": (C.c_int, [C.c_void_p]),
    "grdma_pair_has_pending_writes": (C.c_int, [C.c_void_p]),
    "grdma_pair_readable_size": (C.c_int64, [C.c_void_p]),
    "grdma_pair_writable_size": (C.c_int64, [C.c_void_p]),
    "grdma_pair_state_get": (C.c_int, [C.c_void_p, C.POINTER(PairState)]),
    "grdma_pair_peek_ring": (C.c_int, [C.c_void_p, u64, C.c_void_p, u64]),
    "grdma_pair_peek_staging": (C.c_int, [C.c_void_p, u64, C.c_void_p, u64]),
    "grdma_pair_last_wrs": (C.c_int, [C.c_void_p, C.POINTER((u64 * 2) * 2)]),
    "grdma_pair_ring_device_ptr": (C.c_void_p, [C.c_void_p]),
    "grdma_pair_export_ring_dmabuf": (C.c_int, [C.c_void_p]),
    "grdma_endpoint_write_begin": (C.c_int64, [C.c_void_p, C.POINTER(Slice), u64, C.c_int]),
    "grdma_endpoint_write_step": (C.c_int64, [C.c_void_p, C.POINTER(C.c_int)]),
    "grdma_endpoint_write_abort": (C.c_int, [C.c_void_p]),
    "grdma_endpoint_read": (C.c_int64, [C.c_void_p, u64, C.POINTER(ReadSlice), u64,
                                        C.POINTER(C.c_int)]),
    "grdma_pair_arena_device_ptr": (C.c_void_p, [C.c_void_p]),
    "grdma_pair_arena_size": (u64, [C.c_void_p]),
    "grdma_pair_arena_copy_out": (C.c_int, [C.c_void_p, u64, C.c_void_p, u64]),
    "grdma_poll_pairs": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, u64p,
                                   C.POINTER(C.c_uint8)]),
    "grdma_device_alloc": (C.c_void_p, [u64]),
    "grdma_device_free": (None, [C.c_void_p]),
    "grdma_host_alloc_pinned": (C.c_void_p, [u64]),
    "grdma_host_pin_to_device_node": (C.c_int, []),
    "grdma_host_pin_thread_to_core": (C.c_int, [C.c_int]),
    "grdma_endpoint_write_queue": (C.c_int, [C.c_void_p, C.c_void_p, u64]),
    "grdma_endpoint_write_adopt": (C.c_int, [C.c_void_p]),
    "grdma_endpoint_write_queue_stats": (C.c_int, [C.c_void_p, C.POINTER(u64)]),
    "grdma_endpoint_write_queue_limits": (C.c_int, [C.c_void_p, C.POINTER(u64)]),
    "grdma_endpoint_write_quiesce": (C.c_int, [C.c_void_p]),
    "grdma_host_free_pinned": (None, [C.c_void_p]),
    "grdma_copy_to_device": (C.c_int, [C.c_void_p, C.c_void_p, u64]),
    "grdma_copy_to_host": (C.c_int, [C.c_void_p, C.c_void_p, u64]),
    "grdma_device_synchronize": (C.c_int, []),
    "grdma_pair_pool_reserve": (C.c_int, [C.c_uint32, u64, C.c_int, C.c_int, u64]),
    "grdma_pair_pool_take": (C.c_void_p, [C.c_char_p, u64, C.c_int, C.c_int]),
    "grdma_pair_pool_get": (C.c_void_p, [C.c_char_p]),
    "grdma_pair_pool_putback": (None, [C.c_void_p]),
    "grdma_pair_pool_stats": (C.c_int, [u64p]),
    "grdma_pair_pool_trim": (None, []),
    "grdma_rx_fast_drains": (C.c_int, [u64p]),
    "grdma_rx_table_cache_stats": (C.c_int, [u64p]),
    "grdma_rx_verdict_counts": (C.c_int, [u64p]),
    "grdma_debug_set_promise_wait": (C.c_int, [C.c_uint32]),
    "grdma_tx_fast_sends": (C.c_int, [u64p]),
    "grdma_tx_promise_counts": (C.c_int, [u64p]),
    "grdma_wire_wait_runouts": (u64, []),
    "grdma_stream_job_set_fused_wire": (C.c_int, [C.c_void_p, C.c_int]),
    "grdma_stream_job_wire_groups": (C.c_uint32, [C.c_void_p]),
}

_lib = None


def load():
    """Load the shared object; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GrdmaError(
                "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the HIP data plane has no fallback)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc):
    if rc < 0:
        raise GrdmaError("grdma error %d: %s" % (-rc, load().grdma_last_error().decode()))
    return rc


def init(device=None):
    lib = load()
    if device is None:
        device = int(os.environ.get("GRPC_RDMA_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    check(lib.grdma_init(device))
    return device
