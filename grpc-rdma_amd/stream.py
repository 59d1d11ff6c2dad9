"""Device-resident streaming job (grdma_stream_job_*): pushes a whole slice
list through a connected loop-back link with no host round trips."""
import ctypes as C

from ._lib import GrdmaError, ReadSlice, Slice, check, load

u64 = C.c_uint64


class StreamResult(C.Structure):
    _fields_ = [(n, u64) for n in ("bytes_sent", "bytes_delivered", "slices_delivered",
                                   "tx_rounds", "rx_rounds", "tx_records", "rx_records", "done")] + \
               [("ms_total", C.c_double), ("ms_class", C.c_double * 8),
                ("launches_class", u64 * 8)]


RUN_EAGER, RUN_GRAPH, RUN_INSTRUMENTED, RUN_INSTRUMENTED_SCHEDULE = 0, 1, 2, 4
# classes 5 and 6 exist in RUN_INSTRUMENTED_SCHEDULE only: the launches two stages of neighbouring rounds share
CLASS_NAMES = ["tx_plan", "gather", "wire", "rx_plan", "rx_apply", "plan_pair", "scatter_gather"]

_bound = False


def _bind():
    global _bound
    lib = load()
    if not _bound:
        lib.grdma_stream_job_create.restype = C.c_void_p
        lib.grdma_stream_job_create.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Slice), u64,
                                                C.c_void_p, u64, u64, u64]
        lib.grdma_stream_job_destroy.argtypes = [C.c_void_p]
        lib.grdma_stream_job_run.argtypes = [C.c_void_p, C.c_int, C.POINTER(StreamResult)]
        lib.grdma_stream_job_slices.argtypes = [C.c_void_p, C.POINTER(ReadSlice), u64]
        lib.grdma_stream_job_set_rounds.argtypes = [C.c_void_p, u64]
        lib.grdma_stream_job_set_pipeline.argtypes = [C.c_void_p, C.c_int]
        lib.grdma_stream_job_launch.argtypes = [C.c_void_p]
        lib.grdma_stream_job_launch_streams.argtypes = [C.c_void_p]
        lib.grdma_stream_job_create_multi.restype = C.c_void_p
        lib.grdma_stream_job_create_multi.argtypes = [C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                      C.POINTER(Slice), C.POINTER(u64), C.POINTER(C.c_void_p),
                                                      C.POINTER(u64), C.POINTER(u64), u64]
        lib.grdma_stream_job_slices_of.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(ReadSlice), u64]
        lib.grdma_stream_job_sync.argtypes = [C.c_void_p]
        _bound = True
    return lib


class StreamJob:
    def __init__(self, tx, rx, slices, rx_dst_ptr, rx_dst_cap, slices_cap, max_rounds):
        """slices: list of (device ptr, len)."""
        self.lib = _bind()
        arr = (Slice * len(slices))()
        for i, (p, n) in enumerate(slices):
            arr[i].ptr, arr[i].len = p, n
        self.slices_cap = slices_cap
        self.h = self.lib.grdma_stream_job_create(tx.h, rx.h, arr, len(slices), rx_dst_ptr,
                                                  rx_dst_cap, slices_cap, max_rounds)
        if not self.h:
            raise GrdmaError(self.lib.grdma_last_error().decode())

    def set_rounds(self, n):
        check(self.lib.grdma_stream_job_set_rounds(self.h, n))

    def set_sends(self, sends):
        """`sends` consecutive Sends per round in one plan (grdma_stream_job_set_sends: rdma_flush's loop while the ring
        has room), paired schedule of a pipelined job."""
        self.lib.grdma_stream_job_set_sends.argtypes = [C.c_void_p, C.c_uint32]
        check(self.lib.grdma_stream_job_set_sends(self.h, sends))

    def set_promised_credit(self, on=True):
        """Paired schedule, staged wire: the Send of round t + 1 priced with the credit the drain of round t will post
        (grdma_stream_job_set_promised_credit)."""
        self.lib.grdma_stream_job_set_promised_credit.argtypes = [C.c_void_p, C.c_int]
        check(self.lib.grdma_stream_job_set_promised_credit(self.h, 1 if on else 0))

    def set_fused_wire(self, on=True):
        """Few links, small rings: the wire of a round inside the planner pair's launch (grdma_stream_job_set_fused_wire)."""
        check(self.lib.grdma_stream_job_set_fused_wire(self.h, 1 if on else 0))

    def wire_groups(self):
        """Wire workgroups per link in the planner pair's launch; 0 = the wire is a launch of its own."""
        return int(self.lib.grdma_stream_job_wire_groups(self.h))

    def set_rebuild_index(self, on=True):
        """The slice tables are rewritten between steps: the index the Sends are priced from is rebuilt in every step
        (grdma_stream_job_set_rebuild_index)."""
        self.lib.grdma_stream_job_set_rebuild_index.argtypes = [C.c_void_p, C.c_int]
        check(self.lib.grdma_stream_job_set_rebuild_index(self.h, 1 if on else 0))

    def set_pipeline(self, on):
        check(self.lib.grdma_stream_job_set_pipeline(self.h, 1 if on else 0))

    def run(self, mode=RUN_GRAPH):
        r = StreamResult()
        check(self.lib.grdma_stream_job_run(self.h, mode, C.byref(r)))
        return r

    def launch(self, streams=False):
        if streams:
            check(self.lib.grdma_stream_job_launch_streams(self.h))
        else:
            check(self.lib.grdma_stream_job_launch(self.h))

    def sync(self):
        check(self.lib.grdma_stream_job_sync(self.h))

    def delivered_slices(self, link=0):
        arr = (ReadSlice * self.slices_cap)()
        n = check(self.lib.grdma_stream_job_slices_of(self.h, link, arr, self.slices_cap))
        return [(int(arr[i].off), int(arr[i].len)) for i in range(n)]

    def close(self):
        if self.h:
            self.lib.grdma_stream_job_destroy(self.h)
            self.h = None


class MultiStreamJob(StreamJob):
    """n links advancing in lock step, one op per link in every launch."""

    def __init__(self, links, max_rounds):
        """links: list of (tx Pair, rx Pair, slices [(ptr,len)...], dst ptr, dst cap, slices cap)."""
        self.lib = _bind()
        n = len(links)
        txs = (C.c_void_p * n)(*[l[0].h for l in links])
        rxs = (C.c_void_p * n)(*[l[1].h for l in links])
        total = sum(len(l[2]) for l in links)
        arr = (Slice * total)()
        k = 0
        for l in links:
            for p, ln in l[2]:
                arr[k].ptr, arr[k].len = p, ln
                k += 1
        counts = (u64 * n)(*[len(l[2]) for l in links])
        dsts = (C.c_void_p * n)(*[l[3] for l in links])
        dcaps = (u64 * n)(*[l[4] for l in links])
        scaps = (u64 * n)(*[l[5] for l in links])
        self.slices_cap = max(l[5] for l in links)
        self.h = self.lib.grdma_stream_job_create_multi(n, txs, rxs, arr, counts, dsts, dcaps, scaps,
                                                        max_rounds)
        if not self.h:
            raise GrdmaError(self.lib.grdma_last_error().decode())
