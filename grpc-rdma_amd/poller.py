"""Background poller of the RDMA_BPEV platform (src/core/lib/ibverbs/poller.{h,cc}):
registered pairs get their wakeup fd kicked when they have something for the event
engine.  Thin wrapper over grdma_poller_*: n_threads host threads (GRPC_RDMA_POLLER_THREAD_NUM) sharing one cursor over
the slot table; a pass is host loads of the pairs' pinned state lines (no device call for a pair whose peer lives in
this process; pairs with a remote peer keep one refresh pass of k_poll in flight)."""
import ctypes as C

from ._lib import GrdmaError, check, load


class Poller:
    def __init__(self, n_threads=1, sleep_timeout_ms=1000):
        self.lib = load()
        self.h = self.lib.grdma_poller_create(n_threads, sleep_timeout_ms)
        if not self.h:
            raise GrdmaError(self.lib.grdma_last_error().decode())

    def add(self, pair):
        """AddPollable -> the pair's wakeup fd (add it to an epoll set / poll on it)."""
        return check(self.lib.grdma_poller_add(self.h, pair.h))

    def remove(self, pair):
        check(self.lib.grdma_poller_remove(self.h, pair.h))

    def stats(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(self.lib.grdma_poller_stats(self.h, C.byref(a), C.byref(b)))
        return {"passes": a.value, "wakeups": b.value}

    def close(self):
        if self.h:
            self.lib.grdma_poller_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
