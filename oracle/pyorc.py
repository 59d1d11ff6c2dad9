"""TEST INFRASTRUCTURE ONLY: ctypes bindings for the two CPU checkers.

* ``Oracle``  -> oracle/_build/liboracle.so   (from-scratch C restatement)
* ``RefRing`` -> oracle/_ref/libref_ring.so   (the reference's own
  src/core/lib/ibverbs/ring_buffer.cc, compiled where it lies)

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may
import this module.  The product package never does.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "_build", "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libref_ring.so")
REF_PAIR_TRACE = os.path.join(HERE, "_ref", "ref_pair_trace")
REF_H2_TRACE = os.path.join(HERE, "_ref", "ref_h2_trace")
REF_H2_DEFRAME_TRACE = os.path.join(HERE, "_ref", "ref_h2_deframe_trace")
REF_H2_PERFORM_READ_TRACE = os.path.join(HERE, "_ref", "ref_h2_perform_read_trace")
REF_ENDPOINT_TRACE = os.path.join(HERE, "_ref", "ref_endpoint_trace")

u8p = C.POINTER(C.c_uint8)
u64 = C.c_uint64
u64p = C.POINTER(C.c_uint64)


def build(force=False):
    """Compile the checkers (the reference build only where /root/reference exists)."""
    if force or not os.path.exists(ORACLE_SO) or (
            os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(HERE, "grdma_oracle.c"))):
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if os.path.isdir("/root/reference/src/core/lib/ibverbs") and (
            force or not os.path.exists(REF_SO) or not os.path.exists(REF_PAIR_TRACE) or not os.path.exists(REF_H2_TRACE)
            or os.path.getmtime(REF_H2_TRACE) < os.path.getmtime(os.path.join(HERE, "ref_h2_trace.cc"))
            or not os.path.exists(REF_ENDPOINT_TRACE)
            or os.path.getmtime(REF_ENDPOINT_TRACE) < os.path.getmtime(os.path.join(HERE, "ref_endpoint_trace.cc"))
            or not os.path.exists(REF_H2_PERFORM_READ_TRACE)
            or os.path.getmtime(REF_H2_PERFORM_READ_TRACE) < os.path.getmtime(os.path.join(HERE, "ref_h2_perform_read_trace.cc"))
            or not os.path.exists(REF_H2_DEFRAME_TRACE)
            or os.path.getmtime(REF_H2_DEFRAME_TRACE) < os.path.getmtime(os.path.join(HERE, "ref_h2_deframe_trace.cc"))
            or os.path.getmtime(REF_SO) < os.path.getmtime(os.path.join(HERE, "ref_driver.cc"))
            or os.path.getmtime(REF_PAIR_TRACE) < max(os.path.getmtime(os.path.join(HERE, "ref_pair_trace.cc")),
                                                      os.path.getmtime(os.path.join(HERE, "fakeverbs", "fakeverbs.cc")))):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


# --------------------------------------------------------------------------- C structs
class OrcRing(C.Structure):
    _fields_ = [("buf", C.c_void_p), ("cap", u64), ("mask", u64), ("head", u64),
                ("moving_head", u64), ("remain", u64)]


class OrcSlice(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("len", u64)]


class OrcStatus(C.Structure):
    _fields_ = [("remote_head", u64), ("peer_exit", C.c_int32), ("pad", C.c_int32)]


class OrcPair(C.Structure):
    pass


OrcPair._fields_ = [
    ("ring", OrcRing), ("staging", C.c_void_p), ("staging_cap", u64), ("staging_used", u64),
    ("status_recv", OrcStatus), ("status_send", OrcStatus), ("remote_tail", u64),
    ("internal_read_size", u64), ("partial_write", C.c_int), ("max_sge", C.c_int),
    ("credit_msgs", u64), ("peer", C.POINTER(OrcPair)), ("wr", (u64 * 2) * 2),
    ("wr_count", C.c_int), ("leftover_cap", u64),
    ("zc_buf", C.c_void_p), ("zc_cap", u64), ("zc_tail", u64), ("zc_bytes", u64), ("copy_bytes", u64),
    ("sge_count", u64)]


class OrcEvent(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("a", C.c_uint32), ("b", C.c_uint32),
                ("c", C.c_uint32), ("d", C.c_uint32)]


class OrcStream(C.Structure):
    _fields_ = [("stream_id", C.c_uint32), ("state", C.c_int), ("frame_size", C.c_uint32),
                ("compressed", C.c_int), ("read_closed", C.c_int), ("write_closed", C.c_int),
                ("header_frames_received", C.c_int)]


class OrcParser(C.Structure):
    _fields_ = [("state", C.c_int), ("incoming_frame_size", C.c_uint32),
                ("incoming_frame_type", C.c_uint8), ("incoming_frame_flags", C.c_uint8),
                ("incoming_stream_id", C.c_uint32), ("max_frame_size", C.c_uint32),
                ("check_frame_size", C.c_int), ("cur_parser", C.c_int),
                ("is_server", C.c_int), ("is_first_frame", C.c_int),
                ("expect_continuation_stream_id", C.c_uint32),
                ("header_eof", C.c_int), ("header_boundary", C.c_int),
                ("received_last_frame", C.c_int),
                ("last_new_stream_id", C.c_uint32), ("max_concurrent_streams", C.c_uint32),
                ("streams", C.POINTER(OrcStream)), ("nstreams", u64), ("streams_cap", u64)]


EV_FRAME, EV_PAYLOAD, EV_MSG_BEGIN, EV_MSG_BYTES, EV_MSG_END, EV_STREAM_OPEN, EV_STREAM_CLOSED = 1, 2, 3, 4, 5, 6, 7
H2_SERVER, H2_FIRST_FRAME = 1, 2


def _lib():
    build()
    lib = C.CDLL(ORACLE_SO)
    lib.orc_encoded_size.restype = u64
    lib.orc_encoded_size.argtypes = [u64]
    lib.orc_calc_writable.restype = u64
    lib.orc_calc_writable.argtypes = [u64]
    lib.orc_plan_send.restype = u64
    lib.orc_plan_send.argtypes = [u64, u64, u64, u64, C.c_int, u64p, u64, u64, u64p, u64p]
    lib.orc_pair_init.argtypes = [C.POINTER(OrcPair), u64, C.c_int]
    lib.orc_pair_destroy.argtypes = [C.POINTER(OrcPair)]
    lib.orc_pair_connect.argtypes = [C.POINTER(OrcPair), C.POINTER(OrcPair)]
    lib.orc_pair_send.restype = u64
    lib.orc_pair_send.argtypes = [C.POINTER(OrcPair), C.POINTER(OrcSlice), u64, u64]
    lib.orc_pair_recv.restype = u64
    lib.orc_pair_recv.argtypes = [C.POINTER(OrcPair), C.c_void_p, u64]
    lib.orc_pair_writable.restype = u64
    lib.orc_pair_writable.argtypes = [C.POINTER(OrcPair)]
    lib.orc_pair_enable_zerocopy.argtypes = [C.POINTER(OrcPair), u64]
    lib.orc_pair_allocate_send_buffer.restype = C.c_void_p
    lib.orc_pair_allocate_send_buffer.argtypes = [C.POINTER(OrcPair), u64]
    lib.orc_pair_send_zerocopy.restype = u64
    lib.orc_pair_send_zerocopy.argtypes = [C.POINTER(OrcPair), C.POINTER(OrcSlice), u64, u64]
    lib.orc_endpoint_read.restype = u64
    lib.orc_endpoint_read.argtypes = [C.POINTER(OrcPair), C.c_void_p, u64p]
    lib.orc_ring_readable.restype = u64
    lib.orc_ring_readable.argtypes = [C.POINTER(OrcRing)]
    lib.orc_ring_has_message.argtypes = [C.POINTER(OrcRing)]
    lib.orc_h2_frame_message.restype = C.c_int64
    lib.orc_h2_frame_message.argtypes = [C.c_char_p, u64, C.c_int, C.c_uint32, C.c_uint32,
                                         C.c_int, C.c_void_p, u64, u64p, u64p, u64]
    lib.orc_h2_frame_batch.restype = C.c_int64
    lib.orc_h2_frame_batch.argtypes = [C.POINTER(C.c_char_p), u64p, C.POINTER(C.c_uint32),
                                       C.POINTER(C.c_uint32), u64, C.c_uint32, C.c_void_p, u64,
                                       u64p, u64p, u64]
    lib.orc_h2_parser_init.argtypes = [C.POINTER(OrcParser), C.c_int, C.c_uint32]
    lib.orc_h2_parser_feed.argtypes = [C.POINTER(OrcParser), C.c_char_p, u64,
                                       C.POINTER(OrcEvent), u64, u64p]
    lib.orc_h2_parser_init_ex.argtypes = [C.POINTER(OrcParser), C.c_int, C.c_uint32, C.c_uint32]
    lib.orc_h2_parser_free.argtypes = [C.POINTER(OrcParser)]
    lib.orc_h2_parser_open_stream.argtypes = [C.POINTER(OrcParser), C.c_uint32]
    lib.orc_h2_parser_close_writes.argtypes = [C.POINTER(OrcParser), C.c_uint32]
    lib.orc_h2_parser_live_streams.restype = u64
    lib.orc_h2_parser_live_streams.argtypes = [C.POINTER(OrcParser)]
    lib.orc_stream_baseline.restype = u64
    lib.orc_stream_baseline.argtypes = [u64, C.c_int, C.c_void_p, u64p, u64, u64,
                                        C.POINTER(C.c_double), u64p]
    lib.orc_grpc_msg_header.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
    lib.orc_h2_data_header.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32]
    return lib


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _lib()
    return _LIB


def _keep(slices):
    """-> (ctypes array of (ptr,len), keepalive list)"""
    bufs = [C.create_string_buffer(bytes(s), len(s)) if len(s) else C.create_string_buffer(1)
            for s in slices]
    return bufs


class OracleLink:
    """Two connected oracle pairs (a loop-back connection), side 0 and side 1."""

    def __init__(self, ring_size=4 << 20, max_sge=30):
        self.l = lib()
        self.p = [OrcPair(), OrcPair()]
        for q in self.p:
            assert self.l.orc_pair_init(C.byref(q), ring_size, max_sge) == 0
        self.l.orc_pair_connect(C.byref(self.p[0]), C.byref(self.p[1]))
        self.ring_size = ring_size

    def close(self):
        for q in self.p:
            self.l.orc_pair_destroy(C.byref(q))

    def send(self, side, slices, byte_idx=0):
        bufs = _keep(slices)
        arr = (OrcSlice * max(1, len(slices)))()
        for i, (b, s) in enumerate(zip(bufs, slices)):
            arr[i].ptr = C.addressof(b)
            arr[i].len = len(s)
        return self.l.orc_pair_send(C.byref(self.p[side]), arr, len(slices), byte_idx)

    # ---- zero-copy send buffer (pair.cc:305-323, 793-941).  A slice is bytes, or ("zc", offset,
    # length) for a range of the side's zero-copy buffer.
    def enable_zerocopy(self, side, size):
        assert self.l.orc_pair_enable_zerocopy(C.byref(self.p[side]), size) == 0

    def allocate_send_buffer(self, side, size):
        """-> offset into the zero-copy buffer, or None (AllocateSendBuffer returned nullptr)"""
        q = self.l.orc_pair_allocate_send_buffer(C.byref(self.p[side]), size)
        return None if not q else q - self.p[side].zc_buf

    def zerocopy_write(self, side, off, data):
        C.memmove(self.p[side].zc_buf + off, bytes(data), len(data))

    def send_zerocopy(self, side, slices, byte_idx=0):
        plain = [s for s in slices if not isinstance(s, tuple)]
        bufs = iter(_keep(plain))
        keep = []
        arr = (OrcSlice * max(1, len(slices)))()
        for i, s in enumerate(slices):
            if isinstance(s, tuple):
                arr[i].ptr = self.p[side].zc_buf + s[1]
                arr[i].len = s[2]
            else:
                b = next(bufs)
                keep.append(b)
                arr[i].ptr = C.addressof(b)
                arr[i].len = len(s)
        return self.l.orc_pair_send_zerocopy(C.byref(self.p[side]), arr, len(slices), byte_idx)

    def zerocopy_state(self, side):
        q = self.p[side]
        return dict(tail=q.zc_tail, zerocopy_bytes=q.zc_bytes, copy_bytes=q.copy_bytes, sges=q.sge_count)

    def recv(self, side, cap):
        dst = C.create_string_buffer(max(1, cap))
        n = self.l.orc_pair_recv(C.byref(self.p[side]), dst, cap)
        return dst.raw[:n]

    def endpoint_read(self, side):
        q = self.p[side]
        cap = max(256, self.readable(side), q.leftover_cap)
        dst = C.create_string_buffer(cap)
        alloc = u64(0)
        n = self.l.orc_endpoint_read(C.byref(q), dst, C.byref(alloc))
        return dst.raw[:n], alloc.value

    def readable(self, side):
        return self.l.orc_ring_readable(C.byref(self.p[side].ring))

    def has_message(self, side):
        return bool(self.l.orc_ring_has_message(C.byref(self.p[side].ring)))

    def writable(self, side):
        return self.l.orc_pair_writable(C.byref(self.p[side]))

    def ring_mem(self, side):
        return C.string_at(self.p[side].ring.buf, self.ring_size)

    def staging_mem(self, side):
        q = self.p[side]
        return C.string_at(q.staging, q.staging_used)

    def state(self, side):
        q = self.p[side]
        return dict(head=q.ring.head, moving_head=q.ring.moving_head, remain=q.ring.remain,
                    remote_tail=q.remote_tail, remote_head=q.status_recv.remote_head,
                    internal_read_size=q.internal_read_size, credit_msgs=q.credit_msgs,
                    partial_write=int(q.partial_write))

    def last_wrs(self, side):
        q = self.p[side]
        return [(q.wr[i][0], q.wr[i][1]) for i in range(q.wr_count)]


# --------------------------------------------------------------------------- reference build
def ref_available():
    build()
    return os.path.exists(REF_SO)


_REF = None


def ref():
    global _REF
    if _REF is None:
        build()
        r = C.CDLL(REF_SO)
        for name in ("ref_encoded_size", "ref_calc_writable"):
            getattr(r, name).restype = u64
            getattr(r, name).argtypes = [u64]
        for name in ("ref_reserved_space", "ref_sizeof_grpc_slice",
                     "ref_sizeof_grpc_slice_buffer", "ref_slice_inlined_size"):
            getattr(r, name).restype = u64
        r.ref_link_new.restype = C.c_void_p
        r.ref_link_new.argtypes = [u64, C.c_int]
        r.ref_link_free.argtypes = [C.c_void_p]
        r.ref_pair_send.restype = u64
        r.ref_pair_send.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), u64p, u64, u64,
                                    C.c_int]
        r.ref_pair_recv.restype = u64
        r.ref_pair_recv.argtypes = [C.c_void_p, C.c_int, C.c_void_p, u64]
        r.ref_pair_readable.restype = u64
        r.ref_pair_readable.argtypes = [C.c_void_p, C.c_int]
        r.ref_pair_has_message.argtypes = [C.c_void_p, C.c_int]
        r.ref_pair_writable.restype = u64
        r.ref_pair_writable.argtypes = [C.c_void_p, C.c_int]
        r.ref_pair_ring_mem.restype = C.c_void_p
        r.ref_pair_ring_mem.argtypes = [C.c_void_p, C.c_int]
        r.ref_pair_staging_mem.restype = C.c_void_p
        r.ref_pair_staging_mem.argtypes = [C.c_void_p, C.c_int]
        r.ref_pair_staging_used.restype = u64
        r.ref_pair_staging_used.argtypes = [C.c_void_p, C.c_int]
        r.ref_pair_state.argtypes = [C.c_void_p, C.c_int, u64p]
        r.ref_pair_last_wrs.argtypes = [C.c_void_p, C.c_int, C.POINTER((u64 * 2) * 2)]
        r.ref_pair_enable_zerocopy.argtypes = [C.c_void_p, C.c_int, u64]
        r.ref_pair_allocate_send_buffer.restype = C.c_int64
        r.ref_pair_allocate_send_buffer.argtypes = [C.c_void_p, C.c_int, u64]
        r.ref_pair_zerocopy_mem.restype = C.c_void_p
        r.ref_pair_zerocopy_mem.argtypes = [C.c_void_p, C.c_int]
        r.ref_pair_send_zerocopy.restype = u64
        r.ref_pair_send_zerocopy.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                             u64p, u64, u64]
        r.ref_pair_zerocopy_state.argtypes = [C.c_void_p, C.c_int, u64p]
        r.ref_ring_new.restype = C.c_void_p
        r.ref_ring_new.argtypes = [u64]
        r.ref_ring_free.argtypes = [C.c_void_p]
        r.ref_ring_mem.restype = C.c_void_p
        r.ref_ring_mem.argtypes = [C.c_void_p]
        r.ref_ring_write.restype = u64
        r.ref_ring_write.argtypes = [C.c_void_p, u64, C.c_char_p, u64]
        r.ref_ring_readable.restype = u64
        r.ref_ring_readable.argtypes = [C.c_void_p]
        r.ref_ring_read.restype = u64
        r.ref_ring_read.argtypes = [C.c_void_p, C.c_void_p, u64, u64p]
        r.ref_ring_free_size.restype = u64
        r.ref_ring_free_size.argtypes = [C.c_void_p, u64, u64]
        r.ref_ring_writable.restype = u64
        r.ref_ring_writable.argtypes = [C.c_void_p, u64, u64]
        _REF = r
    return _REF


class RefLink:
    """Same interface as OracleLink, backed by the reference-built ring codec."""

    def __init__(self, ring_size=4 << 20, max_sge=30):
        self.r = ref()
        self.h = self.r.ref_link_new(ring_size, max_sge)
        self.ring_size = ring_size
        self.leftover = [0, 0]

    def close(self):
        self.r.ref_link_free(self.h)

    def send(self, side, slices, byte_idx=0, inline_small=0):
        bufs = _keep(slices)
        n = len(slices)
        ptrs = (C.c_void_p * max(1, n))()
        lens = (u64 * max(1, n))()
        for i, (b, s) in enumerate(zip(bufs, slices)):
            ptrs[i] = C.addressof(b)
            lens[i] = len(s)
        return self.r.ref_pair_send(self.h, side, ptrs, lens, n, byte_idx, inline_small)

    def enable_zerocopy(self, side, size):
        self.r.ref_pair_enable_zerocopy(self.h, side, size)

    def allocate_send_buffer(self, side, size):
        off = self.r.ref_pair_allocate_send_buffer(self.h, side, size)
        return None if off < 0 else off

    def zerocopy_write(self, side, off, data):
        C.memmove(self.r.ref_pair_zerocopy_mem(self.h, side) + off, bytes(data), len(data))

    def send_zerocopy(self, side, slices, byte_idx=0):
        n = len(slices)
        bufs = iter(_keep([s for s in slices if not isinstance(s, tuple)]))
        keep = []
        ptrs = (C.c_void_p * max(1, n))()
        offs = (C.c_int64 * max(1, n))()
        lens = (u64 * max(1, n))()
        for i, s in enumerate(slices):
            if isinstance(s, tuple):
                offs[i], lens[i] = s[1], s[2]
            else:
                b = next(bufs)
                keep.append(b)
                ptrs[i], offs[i], lens[i] = C.addressof(b), -1, len(s)
        return self.r.ref_pair_send_zerocopy(self.h, side, ptrs, offs, lens, n, byte_idx)

    def zerocopy_state(self, side):
        st = (u64 * 4)()
        self.r.ref_pair_zerocopy_state(self.h, side, st)
        return dict(tail=st[0], zerocopy_bytes=st[1], copy_bytes=st[2], sges=st[3])

    def recv(self, side, cap):
        dst = C.create_string_buffer(max(1, cap))
        n = self.r.ref_pair_recv(self.h, side, dst, cap)
        return dst.raw[:n]

    def endpoint_read(self, side):
        """rdma_bp_posix.cc:306-326 + :180-291 driven over the reference ring."""
        readable = self.readable(side)
        alloc = self.leftover[side] or max(256, readable)
        out = b""
        while len(out) < alloc:
            got = self.recv(side, alloc - len(out))
            if not got:
                break
            out += got
        self.leftover[side] = alloc - len(out) if out else alloc
        return out, alloc

    def readable(self, side):
        return self.r.ref_pair_readable(self.h, side)

    def has_message(self, side):
        return bool(self.r.ref_pair_has_message(self.h, side))

    def writable(self, side):
        return self.r.ref_pair_writable(self.h, side)

    def ring_mem(self, side):
        return C.string_at(self.r.ref_pair_ring_mem(self.h, side), self.ring_size)

    def staging_mem(self, side):
        return C.string_at(self.r.ref_pair_staging_mem(self.h, side),
                           self.r.ref_pair_staging_used(self.h, side))

    def state(self, side):
        st = (u64 * 8)()
        self.r.ref_pair_state(self.h, side, st)
        keys = ["head", "moving_head", "remain", "remote_tail", "remote_head",
                "internal_read_size", "credit_msgs", "partial_write"]
        return dict(zip(keys, [int(x) for x in st]))

    def last_wrs(self, side):
        out = ((u64 * 2) * 2)()
        n = self.r.ref_pair_last_wrs(self.h, side, C.byref(out))
        return [(out[i][0], out[i][1]) for i in range(n)]


# --------------------------------------------------------------------------- HTTP/2 helpers
def h2_frame_message(msg, stream_id=1, max_frame=16384, compressed=0, end_stream=0):
    """-> (wire bytes, slice lengths) for one gRPC message (frame_data.cc:64-90)."""
    l = lib()
    n = len(msg)
    frames = (n + 5 + max_frame - 1) // max_frame + 1
    wire_cap = n + 5 + 9 * frames + 64
    wire = C.create_string_buffer(wire_cap)
    lens = (u64 * (3 * frames + 8))()
    wl = u64(0)
    cnt = l.orc_h2_frame_message(bytes(msg), n, compressed, stream_id, max_frame, end_stream,
                                 wire, wire_cap, C.byref(wl), lens, len(lens))
    assert cnt >= 0
    return wire.raw[:wl.value], [int(lens[i]) for i in range(cnt)]


def h2_frame_batch(msgs, stream_ids, flags, max_frame=16384):
    """-> (wire bytes, slice lengths) for messages queued back to back on one outbuf."""
    l = lib()
    n = len(msgs)
    ptrs = (C.c_char_p * n)(*[bytes(m) for m in msgs])
    lens_in = (u64 * n)(*[len(m) for m in msgs])
    sids = (C.c_uint32 * n)(*stream_ids)
    fl = (C.c_uint32 * n)(*flags)
    total = sum(len(m) for m in msgs)
    frames = sum((len(m) + 5 + max_frame - 1) // max_frame + 1 for m in msgs)
    wire_cap = total + 5 * n + 9 * frames + 64
    wire = C.create_string_buffer(wire_cap)
    lens = (u64 * (3 * frames + 8))()
    wl = u64(0)
    cnt = l.orc_h2_frame_batch(ptrs, lens_in, sids, fl, n, max_frame, wire, wire_cap, C.byref(wl),
                               lens, len(lens))
    assert cnt >= 0
    return wire.raw[:wl.value], [int(lens[i]) for i in range(cnt)]


class H2Parser:
    """The deframe side of one chttp2 transport.  expect_client_prefix=True is a fresh server
    connection (preface, SETTINGS first, streams accepted from HEADERS frames); False is a
    client / mid-connection parser whose streams the caller opens (open_stream)."""

    def __init__(self, expect_client_prefix=False, max_frame_size=16384, flags=None,
                 max_concurrent_streams=0xFFFFFFFF):
        self.l = lib()
        self.p = OrcParser()
        if flags is None:
            flags = (H2_SERVER | H2_FIRST_FRAME) if expect_client_prefix else 0
        self.l.orc_h2_parser_init_ex(C.byref(self.p), flags, max_frame_size, max_concurrent_streams)

    def __del__(self):
        try:
            self.l.orc_h2_parser_free(C.byref(self.p))
        except Exception:
            pass

    def open_stream(self, sid):
        return self.l.orc_h2_parser_open_stream(C.byref(self.p), sid)

    def close_writes(self, sid):
        return self.l.orc_h2_parser_close_writes(C.byref(self.p), sid)

    def live_streams(self):
        return int(self.l.orc_h2_parser_live_streams(C.byref(self.p)))

    def feed(self, data, cap=None):
        data = bytes(data)
        cap = cap or (len(data) * 2 + 64)
        ev = (OrcEvent * cap)()
        nev = u64(0)
        rc = self.l.orc_h2_parser_feed(C.byref(self.p), data, len(data), ev, cap, C.byref(nev))
        return rc, [(e.kind, e.a, e.b, e.c, e.d) for e in ev[:nev.value]]


def ref_stream_baseline(ring_size, max_sge, wire, lens, n_msgs):
    """The same pass over the reference-built ring codec (oracle/_ref); -> (payload bytes,
    seconds, checksum).  Raises if oracle/_ref is not there."""
    r = ref()
    r.ref_stream_baseline.restype = u64
    r.ref_stream_baseline.argtypes = [u64, C.c_int, C.c_char_p, C.POINTER(u64), u64, u64,
                                      C.POINTER(C.c_double), C.POINTER(u64)]
    arr = (u64 * len(lens))(*lens)
    buf = C.create_string_buffer(bytes(wire), len(wire))
    sec, chk = C.c_double(0), u64(0)
    n = r.ref_stream_baseline(ring_size, max_sge, buf, arr, len(lens), n_msgs, C.byref(sec), C.byref(chk))
    return n, sec.value, chk.value


def stream_rounds(ring_size, max_sge, wire, lens, passes=1, burst=1):
    """The sequential schedule in C (orc_stream_rounds_burst): `burst` Sends back to back, then the
    reader drains until a read would block, repeat.  -> dict(lens=delivered slice lengths of the last
    pass, rounds=Sends of the first pass that accepted bytes, state={...}, stream_ok, ring_zero)."""
    l = lib()
    l.orc_stream_rounds_burst.argtypes = [u64, C.c_int, C.c_int, C.c_char_p, u64p, u64, C.c_int, u64p, u64, u64p, u64p,
                                          u64p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    arr = (u64 * max(1, len(lens)))(*lens)
    cap = 2 * len(lens) + 64 + sum(lens) // 256
    out = (u64 * cap)()
    n_out, rounds = u64(0), u64(0)
    st = (u64 * 9)()
    ok, zero = C.c_int(0), C.c_int(0)
    rc = l.orc_stream_rounds_burst(ring_size, max_sge, burst, bytes(wire), arr, len(lens), passes, out, cap,
                                   C.byref(n_out), C.byref(rounds), st, C.byref(ok), C.byref(zero))
    assert rc == 0, rc
    names = ["remote_tail", "remote_head", "partial_write", "head", "moving_head", "remain",
             "internal_read_size", "credit_msgs", "leftover_cap"]
    return {"lens": [int(out[i]) for i in range(n_out.value)], "rounds": int(rounds.value),
            "state": {k: int(st[i]) for i, k in enumerate(names)}, "stream_ok": bool(ok.value),
            "ring_zero": bool(zero.value)}


def stream_baseline(ring_size, max_sge, wire, lens, n_msgs, with_checksum=False):
    """Single-thread CPU pass of the full pair protocol; -> (payload bytes, seconds[, checksum])."""
    l = lib()
    arr = (u64 * len(lens))(*lens)
    buf = C.create_string_buffer(bytes(wire), len(wire))
    sec = C.c_double(0)
    chk = u64(0)
    n = l.orc_stream_baseline(ring_size, max_sge, buf, arr, len(lens), n_msgs, C.byref(sec),
                              C.byref(chk))
    if with_checksum:
        return n, sec.value, chk.value
    return n, sec.value
