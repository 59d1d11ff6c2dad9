"""GRPCProfiler mirror (csrc/grdma_stats_time.cc): op names in the reference's order, histogram
statistics, the per-slot table, the opt-in rules of stats_time.cc.  Host only."""
import ctypes as C
import os
import re
import threading

import pytest

REF_HDR = "/root/reference/include/grpcpp/stats_time.h"
# include/grpcpp/stats_time.h:11-44, in order
OPS = ["POLLABLE_EPOLL", "POLLSET_WORK", "TRANSPORT_DO_READ", "TRANSPORT_CONTINUE_READ",
       "TRANSPORT_READ_ALLOCATION_DONE", "TRANSPORT_HANDLE_READ", "TRANSPORT_READ", "TRANSPORT_FLUSH",
       "TRANSPORT_HANDLE_WRITE", "TRANSPORT_WRITE", "PAIR_SEND", "PAIR_RECV", "CLIENT_PREPARE", "CLIENT_CQ_NEXT",
       "SERVER_RPC_REQUEST", "SERVER_RPC_FINISH", "SERVER_CQ_NEXT", "BEGIN_WORKER", "ASYNC_NEXT_INTERNAL",
       "FINALIZE_RESULT", "DESERIALIZE"] + ["ADHOC_%d" % i for i in range(1, 11)]


@pytest.fixture()
def lib(built):
    import grpc_rdma_amd
    L = grpc_rdma_amd.load()
    L.grdma_stats_time_op_name.restype = C.c_char_p
    L.grdma_stats_time_op_name.argtypes = [C.c_int]
    L.grdma_stats_time_add.argtypes = [C.c_int, C.c_int64]
    L.grdma_stats_time_add_custom.argtypes = [C.c_int, C.c_int64]
    L.grdma_stats_time_get.restype = C.c_uint64
    L.grdma_stats_time_get.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.grdma_stats_time_print.restype = C.c_int64
    L.grdma_stats_time_print.argtypes = [C.c_char_p, C.c_uint64]
    L.grdma_stats_time_init.argtypes = [C.c_int]
    L.grdma_stats_time_shutdown()
    yield L
    L.grdma_stats_time_shutdown()


def table(L):
    n = L.grdma_stats_time_print(None, 0)
    buf = C.create_string_buffer(n + 1)
    L.grdma_stats_time_print(buf, n + 1)
    return buf.value.decode()


def get(L, slot, op):
    out = (C.c_double * 5)()
    n = L.grdma_stats_time_get(slot, op, out)
    return n, list(out)


def test_op_names_follow_the_reference_enum(lib):
    names = [lib.grdma_stats_time_op_name(i).decode() for i in range(len(OPS))]
    assert names == OPS
    assert lib.grdma_stats_time_op_name(len(OPS)) == b""
    if os.path.exists(REF_HDR):
        ref = re.findall(r"GRPC_STATS_TIME_([A-Z0-9_]+),", open(REF_HDR).read().split("grpc_stats_time;")[0])
        assert ref[-1] == "MAX_OP_SIZE" and ref[:-1] == OPS
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "grdma_amd.h")).read()
    mine = re.findall(r"GRDMA_STATS_TIME_([A-Z0-9_]+),?\n", hdr.split("typedef enum grdma_stats_time {")[1].split("}")[0])
    assert mine == OPS + ["MAX_OP_SIZE"]


def test_nothing_is_recorded_without_a_slot_or_when_disabled(lib):
    lib.grdma_stats_time_enable()
    lib.grdma_stats_time_add(10, 1000)          # enabled, but this thread has no slot (stats_time.cc:72-80)
    lib.grdma_stats_time_init(0)
    lib.grdma_stats_time_disable()
    lib.grdma_stats_time_add(10, 1000)          # has a slot, but disabled
    assert get(lib, 0, 10)[0] == 0 and lib.grdma_stats_time_enabled() == 0
    lib.grdma_stats_time_enable()
    lib.grdma_stats_time_add(10, 1000)
    assert get(lib, 0, 10)[0] == 1 and lib.grdma_stats_time_enabled() == 1
    lib.grdma_stats_time_add(99, 5)             # out of range: ignored
    lib.grdma_stats_time_add(-1, 5)


def test_histogram_statistics_keep_three_digits(lib):
    lib.grdma_stats_time_init(2)
    lib.grdma_stats_time_enable()
    vals = [(i * 7919) % 100003 + 1 for i in range(20000)] + [5_000_000_000, 123_456_789]
    for v in vals:
        lib.grdma_stats_time_add(11, v)
    n, (mean, p50, p95, p99, mx) = get(lib, 2, 11)
    s = sorted(vals)
    assert n == len(vals) and mx == max(vals)
    assert abs(mean - sum(vals) / len(vals)) <= 1e-6 * mean
    for got, q in ((p50, 0.5), (p95, 0.95), (p99, 0.99)):
        exact = s[max(0, int(q * len(s) + 0.5) - 1)]
        assert abs(got - exact) <= 1.5e-3 * exact, (q, got, exact)
    # small values are exact
    lib.grdma_stats_time_init(3)
    for v in (3, 3, 3, 2047, 2047, 9):
        lib.grdma_stats_time_add(0, v)
    assert get(lib, 3, 0) == (6, [pytest.approx(685.3333333), 3.0, 2047.0, 2047.0, 2047.0])


def test_table_has_the_reference_columns_and_units(lib, monkeypatch):
    lib.grdma_stats_time_init(1)
    lib.grdma_stats_time_enable()
    for v in (2000, 4000, 6000):
        lib.grdma_stats_time_add(9, v)           # TRANSPORT_WRITE, nanoseconds
    lib.grdma_stats_time_add_custom(21, 42)       # ADHOC_1, a custom quantity
    monkeypatch.delenv("GRPC_PROFILING_UNIT", raising=False)
    t = table(lib)
    assert "Profiling Result" in t and "Unit us" in t and "Slot: 1" in t
    assert re.search(r"Name\s*\|\s*Count\s*\|\s*Mean\s*\|\s*P50\s*\|\s*P95\s*\|\s*P99\s*\|\s*MAX", t)
    row = [ln for ln in t.splitlines() if "TRANSPORT_WRITE" in ln][0]
    cells = [c.strip() for c in row.strip("|").split("|")]
    # mean and percentiles in us, MAX divided by the unit scale only (stats_time.cc:214-225)
    assert cells[1:] == ["3", "4.00", "4.00", "6.00", "6.00", "6000.00"]
    assert "ADHOC_1 (custom)" in t and "42.00" in t
    monkeypatch.setenv("GRPC_PROFILING_UNIT", "milli")
    t = table(lib)
    assert "Unit ms" in t
    row = [ln for ln in t.splitlines() if "TRANSPORT_WRITE" in ln][0]
    assert [c.strip() for c in row.strip("|").split("|")][2] == "0.00"


def test_slots_belong_to_threads(lib):
    lib.grdma_stats_time_enable()

    def worker(slot, n):
        lib.grdma_stats_time_init(slot)
        for i in range(n):
            lib.grdma_stats_time_add(6, 100 + slot)

    ts = [threading.Thread(target=worker, args=(s, 50 * (s + 1))) for s in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for s in range(4):
        n, st = get(lib, s, 6)
        assert n == 50 * (s + 1) and st[4] == 100 + s
