"""GPU test of the background poller (RDMA_BPEV): the behaviour of
src/core/lib/ibverbs/poller.cc:52-106 as the event engine sees it -- a pair's wakeup fd
becomes readable when the pair has a message, is not signalled twice before the consumer
read it, keeps being signalled while the message is unread (level triggered, the engine
re-arms), goes quiet after the drain, and fires for a half-closed peer."""
import select
import time

import pytest

pytestmark = pytest.mark.gpu


def _readable(fd, timeout):
    r, _, _ = select.select([fd], [], [], timeout)
    return bool(r)


def test_poller_wakes_the_right_pair(gpu):
    g = gpu
    from grpc_rdma_amd.poller import Poller
    links = []
    for _ in range(3):
        a, b = g.Pair(1 << 18, 30), g.Pair(1 << 18, 30)
        g.connect_pairs(a, b)
        links.append((a, b))
    pl = Poller(1, 50)
    fds = [pl.add(b) for _, b in links]
    assert len(set(fds)) == 3 and all(fd >= 0 for fd in fds)
    time.sleep(0.1)
    assert not any(_readable(fd, 0) for fd in fds), "idle pairs must not be woken"
    # a message for link 1 only
    payload = bytes(range(200)) * 3
    assert links[1][0].Send([payload]) == len(payload)
    assert _readable(fds[1], 2.0), "the poller did not signal the pair that has a message"
    assert not _readable(fds[0], 0) and not _readable(fds[2], 0)
    w0 = pl.stats()["wakeups"]
    time.sleep(0.1)
    assert pl.stats()["wakeups"] == w0, "signalled again although the wakeup was not consumed (poller.cc:76-78)"
    # consume without reading the message: the pair still has it, so it is signalled again
    assert links[1][1].lib.grdma_pair_consume_wakeup(links[1][1].h) == 1
    assert _readable(fds[1], 2.0)
    # drain, consume: quiet afterwards
    got, _ = links[1][1].endpoint_read(8)
    assert b"".join(got) == payload
    links[1][1].lib.grdma_pair_consume_wakeup(links[1][1].h)
    time.sleep(0.1)
    # (one more wakeup may have been written between the drain and the consume)
    links[1][1].lib.grdma_pair_consume_wakeup(links[1][1].h)
    time.sleep(0.1)
    assert not _readable(fds[1], 0), "woken with nothing to read"
    # peer exit: Disconnect() on one side half-closes the other (pair.cc:325-347)
    links[2][0].Disconnect()
    assert _readable(fds[2], 2.0), "a half-closed pair must wake the engine (poller.cc:89-93)"
    st = pl.stats()
    assert st["passes"] > 0 and st["wakeups"] >= 3
    for _, b in links:
        pl.remove(b)
    pl.close()
    for a, b in links:
        a.close()
        b.close()


def test_poller_rejects_bad_arguments(gpu):
    from grpc_rdma_amd.poller import Poller
    from grpc_rdma_amd._lib import GrdmaError
    with pytest.raises(GrdmaError):
        Poller(0, 10)  # GRPC_RDMA_POLLER_THREAD_NUM must be positive (config.cc)
    pl = Poller(1, 10)
    a = gpu.Pair(1 << 16, 30)
    with pytest.raises(GrdmaError):
        pl.remove(a)  # never added
    pl.close()
    a.close()
