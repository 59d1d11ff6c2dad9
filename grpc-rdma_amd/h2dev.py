"""Device-side HTTP/2 DATA framing / deframing (grdma_h2_*), thin ctypes wrappers."""
import ctypes as C

from ._lib import GrdmaError, ReadSlice, Slice, check, load

u64 = C.c_uint64


class H2Msg(C.Structure):
    _fields_ = [("payload", C.c_void_p), ("len", u64), ("stream_id", C.c_uint32),
                ("flags", C.c_uint32)]


class H2Event(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("kind", "a", "b", "c", "d", "slice")]


_bound = False


def _bind():
    global _bound
    lib = load()
    if not _bound:
        lib.grdma_h2_frame_messages.restype = C.c_int64
        lib.grdma_h2_frame_messages.argtypes = [C.POINTER(H2Msg), u64, C.c_uint32, C.c_void_p, u64,
                                                C.c_void_p, u64, C.POINTER(u64)]
        lib.grdma_h2_parser_create.restype = C.c_void_p
        lib.grdma_h2_parser_create.argtypes = [C.c_int, C.c_uint32]
        lib.grdma_h2_parser_destroy.argtypes = [C.c_void_p]
        lib.grdma_h2_parser_create_ex.restype = C.c_void_p
        lib.grdma_h2_parser_create_ex.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32]
        for fn in (lib.grdma_h2_parser_open_streams, lib.grdma_h2_parser_close_writes):
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]
        lib.grdma_h2_parser_live_streams.restype = C.c_int64
        lib.grdma_h2_parser_live_streams.argtypes = [C.c_void_p]
        lib.grdma_h2_last_boundary_steps.restype = C.c_uint64
        lib.grdma_h2_last_boundary_steps.argtypes = []
        lib.grdma_h2_parser_chunk_stats.restype = C.c_int
        lib.grdma_h2_parser_chunk_stats.argtypes = [C.c_void_p, C.POINTER(u64)]
        lib.grdma_h2_deframe.restype = C.c_int64
        lib.grdma_h2_deframe.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(ReadSlice), u64,
                                         C.POINTER(H2Event), u64, C.POINTER(C.c_int)]
        _bound = True
    return lib


def frame_messages(msgs, max_frame, slices_dev_ptr, slices_cap, hdr_dev_ptr, hdr_cap):
    """msgs: list of (payload device ptr, len, stream_id, flags). -> (nslices, wire_bytes)."""
    lib = _bind()
    arr = (H2Msg * len(msgs))()
    for i, (p, n, sid, fl) in enumerate(msgs):
        arr[i].payload, arr[i].len, arr[i].stream_id, arr[i].flags = p, n, sid, fl
    wire = u64(0)
    n = check(lib.grdma_h2_frame_messages(arr, len(msgs), max_frame, slices_dev_ptr, slices_cap,
                                          hdr_dev_ptr, hdr_cap, C.byref(wire)))
    return n, wire.value


H2_SERVER, H2_FIRST_FRAME, H2_BOUNDARY_STEP, H2_NO_BOUNDARY_STEP, H2_BULK_PAIRS, H2_TICKS, H2_NO_BULK_PAIRS = 1, 2, 4, 8, 16, 32, 64
H2_NO_CHUNKS = 128


class Parser:
    """Deframe state + stream map of one transport in device memory (grdma_h2_parser).
    expect_client_prefix=True: a fresh server connection (streams accepted from HEADERS);
    False: a client / mid-connection parser whose streams the caller opens."""

    def __init__(self, expect_client_prefix=False, max_frame_size=16384, flags=None,
                 max_concurrent_streams=0xFFFFFFFF, table_slots=0, boundary_step=None, bulk_pairs=None, ticks=False, chunks=None):
        self.lib = _bind()
        if flags is None:
            flags = (H2_SERVER | H2_FIRST_FRAME) if expect_client_prefix else 0
        if boundary_step is not None:  # None: GRDMA_H2_BOUNDARY_STEP in the environment decides
            flags |= H2_BOUNDARY_STEP if boundary_step else H2_NO_BOUNDARY_STEP
        if bulk_pairs is not None:  # None: the library default (64 frames per bulk step unless GRDMA_H2_BULK_PAIRS=0)
            flags |= H2_BULK_PAIRS if bulk_pairs else H2_NO_BULK_PAIRS
        if ticks:
            flags |= H2_TICKS
        if chunks is not None and not chunks:  # None / True: the library default (GRDMA_H2_CHUNKS, lists of >= 2048 slices)
            flags |= H2_NO_CHUNKS
        self.h = self.lib.grdma_h2_parser_create_ex(flags, max_frame_size, max_concurrent_streams, table_slots)
        if not self.h:
            raise GrdmaError("h2 parser allocation failed")

    def _ids(self, ids):
        ids = list(ids)
        return (C.c_uint32 * max(1, len(ids)))(*ids), len(ids)

    def open_streams(self, ids):
        arr, n = self._ids(ids)
        return check(self.lib.grdma_h2_parser_open_streams(self.h, arr, n))

    def close_writes(self, ids):
        arr, n = self._ids(ids)
        return check(self.lib.grdma_h2_parser_close_writes(self.h, arr, n))

    def live_streams(self):
        return check(self.lib.grdma_h2_parser_live_streams(self.h))

    def chunk_stats(self):
        """(calls the chunked deframer planned, calls whose chunks verified and were merged)"""
        out = (u64 * 2)()
        check(self.lib.grdma_h2_parser_chunk_stats(self.h, out))
        return int(out[0]), int(out[1])

    def chunk_phases(self, kmax=256):
        """profiling aid: per chunk (start, cuts found, map copied, parsed, compared, slices) and the merge's stamps, in device-clock ticks relative to the earliest"""
        n = (kmax + 1) * 8
        out = (u64 * n)()
        self.lib.grdma_h2_parser_chunk_dbg.restype = C.c_int
        self.lib.grdma_h2_parser_chunk_dbg.argtypes = [C.c_void_p, C.POINTER(u64), u64]
        check(self.lib.grdma_h2_parser_chunk_dbg(self.h, out, n))
        rows = [[int(out[r * 8 + c]) for c in range(8)] for r in range(kmax + 1)]
        return rows

    def deframe(self, arena_dev_ptr, slices, cap=None):
        """slices: list of (offset, len) in the arena. -> (h2 error, events)"""
        n = len(slices)
        arr = (ReadSlice * max(1, n))()
        for i, (o, l) in enumerate(slices):
            arr[i].off, arr[i].len = o, l
        cap = cap or (sum(l for _, l in slices) * 2 + 64 if n else 64)
        cap = min(cap, 1 << 20)
        ev = (H2Event * cap)()
        err = C.c_int(0)
        m = check(self.lib.grdma_h2_deframe(self.h, arena_dev_ptr, arr, n, ev, cap, C.byref(err)))
        self.last_boundary_steps = int(self.lib.grdma_h2_last_boundary_steps())
        return err.value, [(e.kind, e.a, e.b, e.c, e.d, e.slice) for e in ev[:m]]

    def close(self):
        if self.h:
            self.lib.grdma_h2_parser_destroy(self.h)
            self.h = None


class Pipe:
    """frame -> streaming job -> deframe as one enqueued device pipeline (grdma_h2_pipe).
    msgs: list of (payload device ptr, len, stream_id, flags); the job must have been run once."""

    def __init__(self, job, msgs, parser, delivered_slices, events_cap, link=0, max_frame=16384):
        self.lib = _bind()
        lib = self.lib
        lib.grdma_h2_pipe_create.restype = C.c_void_p
        lib.grdma_h2_pipe_create.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(H2Msg), u64, C.c_uint32, C.c_void_p, u64, u64]
        lib.grdma_h2_pipe_enqueue.argtypes = [C.c_void_p, C.c_int]
        lib.grdma_h2_pipe_sync.argtypes = [C.c_void_p, C.POINTER(u64), C.POINTER(H2Event), u64]
        lib.grdma_h2_pipe_destroy.argtypes = [C.c_void_p]
        lib.grdma_h2_pipe_boundary_stats.argtypes = [C.c_void_p, C.POINTER(u64)]
        arr = (H2Msg * len(msgs))()
        for i, (p, n, sid, fl) in enumerate(msgs):
            arr[i].payload, arr[i].len, arr[i].stream_id, arr[i].flags = p, n, sid, fl
        self.events_cap = events_cap
        self.delivered = delivered_slices
        self.parser = parser  # (kept alive)
        self.h = lib.grdma_h2_pipe_create(job.h, link, arr, len(msgs), max_frame, parser.h, delivered_slices, events_cap)
        if not self.h:
            raise GrdmaError("h2 pipe allocation failed")

    def enqueue(self, _unused=False):
        check(self.lib.grdma_h2_pipe_enqueue(self.h, 0))

    def sync(self, want_events=False):
        """-> dict(framed, frame_overflow, events, deframe_overflow, parsed, h2_error[, event list])"""
        out = (u64 * 14)()
        ev = (H2Event * self.events_cap)() if want_events else None
        check(self.lib.grdma_h2_pipe_sync(self.h, out, ev, self.events_cap if want_events else 0))
        r = dict(zip(("framed", "frame_overflow", "events", "deframe_overflow", "parsed", "h2_error", "frame_us", "deframe_us", "bulk_steps", "bulk_frames", "t_wait", "t_bulk", "t_total", "t_serial"),
                     [int(x) for x in out]))
        bs = (u64 * 2)()
        check(self.lib.grdma_h2_pipe_boundary_stats(self.h, bs))
        r["boundary_steps"], r["t_boundary"] = int(bs[0]), int(bs[1])
        if want_events:
            r["event_list"] = [(e.kind, e.a, e.b, e.c, e.d, e.slice) for e in ev[:min(r["events"], self.events_cap)]]
        return r

    def close(self):
        if self.h:
            self.lib.grdma_h2_pipe_destroy(self.h)
            self.h = None
