"""Host-side mirror of PairPollable (src/core/lib/ibverbs/pair.h:106-152) and of
the endpoint read/write loops (src/core/lib/iomgr/rdma_bp_posix.cc), as thin
wrappers over the C ABI.  Same method names and argument meaning as the
reference so the parity tests read like pair-level tests of the reference."""
import ctypes as C

from . import _lib
from ._lib import GrdmaError, PairState, ReadSlice, Slice, check, load

MEM_DEVICE, MEM_HOST = 0, 1
WIRE_STAGED, WIRE_DIRECT = 0, 2
RING_FINE_GRAINED, WIRE_ORDERED = 4, 8


class DeviceBuffer:
    """A byte buffer in HBM (payloads that are already resident on the GPU)."""

    def __init__(self, nbytes=None, data=None, offset=0):
        lib = load()
        if data is not None:
            nbytes = len(data)
        self.nbytes = nbytes
        self.offset = offset  # lets tests place payloads at odd alignments
        self._base = lib.grdma_device_alloc(nbytes + offset + 64)
        if not self._base:
            raise GrdmaError(lib.grdma_last_error().decode())
        self.ptr = self._base + offset
        if data is not None and nbytes:
            src = C.create_string_buffer(bytes(data), nbytes)
            check(lib.grdma_copy_to_device(self.ptr, src, nbytes))

    def read(self, n=None, off=0):
        n = self.nbytes - off if n is None else n
        dst = C.create_string_buffer(max(1, n))
        if n:
            check(load().grdma_copy_to_host(dst, self.ptr + off, n))
        return dst.raw[:n]

    def free(self):
        if self._base:
            load().grdma_device_free(self._base)
            self._base = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


ZC_MEM_HOST, ZC_MEM_BAR, ZC_MEM_DEVICE = 0, 1, 2


class Pair:
    def __init__(self, ring_size=4 << 20, max_sge=30, flags=WIRE_STAGED, handle=None):
        self.lib = load()
        self.ring_size = ring_size
        # handle: wrap a pair somebody else owns (grdma_pair_pool_take); detach() before it goes back
        self.h = handle if handle is not None else self.lib.grdma_pair_create(ring_size, max_sge, flags)
        if not self.h:
            raise GrdmaError(self.lib.grdma_last_error().decode())

    def detach(self):
        """Forget the handle without destroying the pair (it belongs to the PairPool)."""
        self.h = None

    def close(self):
        if self.h:
            self.lib.grdma_pair_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- PairPollable ------------------------------------------------------------
    @staticmethod
    def _slices(slices):
        """slices: list of (ptr, len) or DeviceBuffer or bytes (host memory)."""
        arr = (Slice * max(1, len(slices)))()
        keep, host = [], False
        for i, s in enumerate(slices):
            if isinstance(s, DeviceBuffer):
                arr[i].ptr, arr[i].len = s.ptr, s.nbytes
            elif isinstance(s, (bytes, bytearray)):
                b = C.create_string_buffer(bytes(s), len(s)) if len(s) else C.create_string_buffer(1)
                keep.append(b)
                arr[i].ptr, arr[i].len = C.addressof(b), len(s)
                host = True
            else:
                arr[i].ptr, arr[i].len = s
        return arr, keep, host

    def Send(self, slices, byte_idx=0, flags=None):
        arr, keep, host = self._slices(slices)
        if flags is None:
            flags = MEM_HOST if host else MEM_DEVICE
        return check(self.lib.grdma_pair_send(self.h, arr, len(slices), byte_idx, flags))

    # -- zero-copy send buffer (pair.cc:305-323, 793-941) ----------------------------
    def enable_zerocopy(self, nbytes=0, mem=None):
        """mem: None = GRPC_RDMA_HIP_ZEROCOPY_MEM (default "host"), or ZC_MEM_HOST / ZC_MEM_BAR / ZC_MEM_DEVICE."""
        if mem is None:
            check(self.lib.grdma_pair_enable_zerocopy(self.h, nbytes))
        else:
            check(self.lib.grdma_pair_enable_zerocopy_ex(self.h, nbytes, mem))

    def zerocopy_mem(self):
        return self.lib.grdma_pair_zerocopy_mem(self.h)

    def AllocateSendBuffer(self, size):
        """-> pointer into the pair's zero-copy buffer (host-writable unless the buffer is ZC_MEM_DEVICE), or None
        (as the reference's nullptr)."""
        return self.lib.grdma_pair_allocate_send_buffer(self.h, size) or None

    def SendZerocopy(self, slices, byte_idx=0):
        """slices: DeviceBuffer or (ptr, len) -- ranges of the zero-copy buffer among them -- or, with a host-writable
        zero-copy buffer, bytes (host memory: what grpc_endpoint_write holds beside the serialised message); a list
        with host slices is sent with GRDMA_MEM_HOST, so everything in it that is not a range of the zero-copy buffer
        must be host memory."""
        arr, keep, host = self._slices(slices)
        if host and any(isinstance(s, DeviceBuffer) for s in slices):
            raise GrdmaError("SendZerocopy: host slices and device buffers cannot be mixed in one call")
        return check(self.lib.grdma_pair_send_zerocopy(self.h, arr, len(slices), byte_idx, MEM_HOST if host else MEM_DEVICE))

    def zerocopy_state(self):
        out = (C.c_uint64 * 4)()
        check(self.lib.grdma_pair_zerocopy_state(self.h, out))
        return dict(tail=out[0], zerocopy_bytes=out[1], copy_bytes=out[2], sges=out[3])

    def Recv(self, capacity):
        """-> bytes (copied to the host for inspection)."""
        dst = C.create_string_buffer(max(1, capacity))
        n = check(self.lib.grdma_pair_recv(self.h, dst, capacity, MEM_HOST))
        return dst.raw[:n]

    def HasMessage(self):
        return bool(check(self.lib.grdma_pair_has_message(self.h)))

    def HasPendingWrites(self):
        return bool(check(self.lib.grdma_pair_has_pending_writes(self.h)))

    def GetReadableSize(self):
        return check(self.lib.grdma_pair_readable_size(self.h))

    def GetWritableSize(self):
        return check(self.lib.grdma_pair_writable_size(self.h))

    def get_status(self):
        return check(self.lib.grdma_pair_get_status(self.h))

    # -- bootstrap with a peer in another process (rdma_bp_posix.cc:763-784) ------------------
    BLOB_BYTES = 48 + 32 + 128

    def export_address(self):
        """-> the bootstrap blob (first 48 bytes = the reference's Address, address.h:24-31)."""
        buf = C.create_string_buffer(self.BLOB_BYTES)
        check(self.lib.grdma_pair_export_address(self.h, buf))
        return buf.raw

    def connect_remote(self, blob):
        buf = C.create_string_buffer(bytes(blob), self.BLOB_BYTES)
        check(self.lib.grdma_pair_connect_remote(self.h, buf))

    def bootstrap_fd(self, fd):
        """exchange_data over a connected socket + Connect()."""
        check(self.lib.grdma_pair_bootstrap_fd(self.h, fd))

    def Disconnect(self):
        check(self.lib.grdma_pair_disconnect(self.h))

    # -- endpoint ------------------------------------------------------------------
    def endpoint_write(self, slices, flags=None):
        """rdma_write + the rdma_flush retries.  -> list of per-step sent counts;
        stops early (list ends with 0) when no credit is left."""
        arr, keep, host = self._slices(slices)
        if flags is None:
            flags = MEM_HOST if host else MEM_DEVICE
        check(self.lib.grdma_endpoint_write_begin(self.h, arr, len(slices), flags))
        # the pair reads the caller's buffers until the write is done (the contract of
        # grpc_endpoint_write, endpoint.h:78-91): keep them alive across the continuation calls
        self._w_keep = (arr, keep)
        return self.endpoint_write_continue()

    def endpoint_write_continue(self):
        steps, done = [], C.c_int(0)
        while not done.value:
            n = check(self.lib.grdma_endpoint_write_step(self.h, C.byref(done)))
            steps.append(n)
            if n == 0:
                break
        if done.value:
            self._w_keep = None
        return steps, bool(done.value)

    def endpoint_read(self, max_reads=1):
        """-> (list of bytes, one per endpoint_read completion, would_block)."""
        cap = min(max_reads, 8192)
        arr = (ReadSlice * cap)()
        wb = C.c_int(0)
        n = check(self.lib.grdma_endpoint_read(self.h, max_reads, arr, cap, C.byref(wb)))
        out = []
        for i in range(n):
            dst = C.create_string_buffer(max(1, arr[i].len))
            check(self.lib.grdma_pair_arena_copy_out(self.h, arr[i].off, dst, arr[i].len))
            out.append(dst.raw[:arr[i].len])
        return out, bool(wb.value)

    def set_latency_mode(self, on=True):
        self.lib.grdma_pair_set_latency_mode.argtypes = [C.c_void_p, C.c_int]
        check(self.lib.grdma_pair_set_latency_mode(self.h, int(on)))

    def arm_read(self, max_reads=64):
        """grdma_pair_arm_read: a standing read order, carried out by a watcher workgroup of the latency engine when
        bytes land in this pair's ring; 0 disarms."""
        self.lib.grdma_pair_arm_read.argtypes = [C.c_void_p, C.c_uint64]
        check(self.lib.grdma_pair_arm_read(self.h, int(max_reads)))

    def watch_hits(self):
        """Completions a watcher workgroup of the engine produced (arrival-triggered drains) that a read has taken."""
        self.lib.grdma_pair_watch_hits.argtypes = [C.c_void_p]
        self.lib.grdma_pair_watch_hits.restype = C.c_int64
        return int(self.lib.grdma_pair_watch_hits(self.h))

    def armed_ready(self):
        self.lib.grdma_pair_armed_ready.argtypes = [C.c_void_p]
        return int(self.lib.grdma_pair_armed_ready(self.h))

    # -- observability -------------------------------------------------------------
    def state(self):
        st = PairState()
        check(self.lib.grdma_pair_state_get(self.h, C.byref(st)))
        return {n: int(getattr(st, n)) for n, _ in PairState._fields_}

    def ring_mem(self):
        dst = C.create_string_buffer(self.ring_size)
        check(self.lib.grdma_pair_peek_ring(self.h, 0, dst, self.ring_size))
        return dst.raw

    def staging_mem(self, n):
        dst = C.create_string_buffer(max(1, n))
        if n:
            check(self.lib.grdma_pair_peek_staging(self.h, 0, dst, n))
        return dst.raw[:n]

    def last_wrs(self):
        out = ((C.c_uint64 * 2) * 2)()
        n = check(self.lib.grdma_pair_last_wrs(self.h, C.byref(out)))
        return [(int(out[i][0]), int(out[i][1])) for i in range(n)]


def connect_pairs(a, b):
    check(load().grdma_pair_connect(a.h, b.h))


def poll_pairs(pairs):
    """Batched HasMessage/GetReadableSize: -> (readable list, has_message list)."""
    n = len(pairs)
    hs = (C.c_void_p * n)(*[p.h for p in pairs])
    rd = (C.c_uint64 * n)()
    hm = (C.c_uint8 * n)()
    check(load().grdma_poll_pairs(hs, n, rd, hm))
    return [int(x) for x in rd], [bool(x) for x in hm]


def pingpong(a, b, req_slices, resp_slices, iters=1000, warmup=100):
    """Unary ping-pong a -> b -> a (grdma_pingpong).  Slices are host bytes.
    -> (list of RTT in ns, [phase ns sums: client write, server read, server write, client read])"""
    lib = load()
    lib.grdma_pingpong.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Slice), C.c_uint64,
                                   C.POINTER(Slice), C.c_uint64, C.c_int, C.c_uint64, C.c_uint64,
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    ra, ka, _ = Pair._slices(req_slices)
    rb, kb, _ = Pair._slices(resp_slices)
    rtt = (C.c_uint64 * iters)()
    ph = (C.c_uint64 * 4)()
    check(lib.grdma_pingpong(a.h, b.h, ra, len(req_slices), rb, len(resp_slices), MEM_HOST, iters,
                             warmup, rtt, ph))
    return [int(x) for x in rtt], [int(x) for x in ph]
