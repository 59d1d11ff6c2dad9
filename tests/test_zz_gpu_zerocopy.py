"""GPU parity: grdma_pair_allocate_send_buffer / grdma_pair_send_zerocopy (k_tx_plan_zc + k_copy) against
the CPU oracle's AllocateSendBuffer / SendZerocopy (pair.cc:305-323, 793-941; the oracle is pinned against the
reference-built pair.cc itself in tests/test_oracle_vs_ref.py).

Round 6: the buffer is HOST-WRITABLE by default (pinned mapped host memory the gather reads in place; GRDMA_ZC_MEM_*),
because the reference's caller serialises into it with the CPU (GenericSerialize -> SerializeWithCachedSizesToArray,
include/grpcpp/impl/codegen/proto_utils.h:68-95).  The tests below write it the way that caller does -- plain CPU
stores through the returned pointer (ctypes.memmove) -- and keep a device-memory variant for device-side serialisers."""
import ctypes as C
import os
import random

import pytest

from oracle import pyorc
from tests.test_gpu_pair_parity import _ring_eq, check_state, mk_link

pytestmark = pytest.mark.gpu


def zc_write(g, pair, ptr, data):
    """The caller's serialisation: CPU stores through the pointer AllocateSendBuffer returned (a device-only buffer is
    filled with a copy instead)."""
    from grpc_rdma_amd.pair import ZC_MEM_DEVICE
    if pair.zerocopy_mem() == ZC_MEM_DEVICE:
        g._lib.check(g.load().grdma_copy_to_device(ptr, data, len(data)))
    else:
        C.memmove(ptr, bytes(data), len(data))


@pytest.mark.parametrize("mem", [0, 2], ids=["host_writable", "device_memory"])
@pytest.mark.parametrize("seed", range(8))
def test_zerocopy_sequences_match_oracle(gpu, seed, mem):
    g = gpu
    if mem == 2 and seed >= 3:
        pytest.skip("three seeds cover the device-memory variant")
    rng = random.Random(500 + seed)
    R = rng.choice([64, 256, 4096, 65536])
    sge = rng.choice([3, 4, 5, 8, 30, 100])
    Z = rng.choice([64, 4096, 2 * R])
    a, b = mk_link(g, R, sge)
    o = pyorc.OracleLink(R, sge)
    a.enable_zerocopy(Z, mem=mem)
    assert a.zerocopy_mem() == mem
    o.enable_zerocopy(0, Z)
    zc_base = None
    sizes = [1, 2, 7, 8, 9, 15, 16, 17, 23, 24, 100, 255, 256, 257, R // 3, R, Z // 2]
    for step in range(40):
        op = rng.random()
        if op < 0.55:
            dsl, osl, keep = [], [], []
            for _ in range(rng.randint(1, 5)):
                n = max(1, rng.choice(sizes))
                data = bytes(rng.getrandbits(8) for _ in range(n))
                off = None
                if rng.random() < 0.5 and n <= Z:
                    dptr, ooff = a.AllocateSendBuffer(n), o.allocate_send_buffer(0, n)
                    assert (dptr is None) == (ooff is None)
                    if dptr is not None:
                        zc_base = zc_base or dptr
                        off = dptr - zc_base
                        assert off == ooff
                    elif zc_base is not None and rng.random() < 0.5:
                        off = rng.randrange(0, Z - n + 1)
                if off is not None:
                    zc_write(g, a, zc_base + off, data)
                    o.zerocopy_write(0, off, data)
                    dsl.append((zc_base + off, n))
                    osl.append(("zc", off, n))
                else:
                    buf = g.DeviceBuffer(data=data, offset=rng.randrange(16))
                    keep.append(buf)
                    dsl.append(buf)
                    osl.append(data)
            first = osl[0][2] if isinstance(osl[0], tuple) else len(osl[0])
            bi = rng.randrange(first) if rng.random() < 0.3 else 0
            assert a.SendZerocopy(dsl, bi) == o.send_zerocopy(0, osl, bi), (seed, step)
            assert a.last_wrs() == o.last_wrs(0)
            assert a.zerocopy_state() == o.zerocopy_state(0)
        elif op < 0.65:
            data = bytes(rng.getrandbits(8) for _ in range(rng.choice(sizes[:14])))
            assert a.Send([g.DeviceBuffer(data=data)]) == o.send(0, [data])
        else:
            cap = rng.choice([1, 8, 64, 256, R])
            assert b.Recv(cap) == o.recv(1, cap)
        assert _ring_eq(b.ring_mem(), o.ring_mem(1)), (seed, step)
        check_state(a, b, o)
    a.close(); b.close(); o.close()


def test_zerocopy_rules(gpu):
    """One allocation at a time; a zero-copy record is limited by the receiver's credit, not by the staging
    buffer; host slices are refused."""
    g = gpu
    R = 4096
    a, b = mk_link(g, R, 30)
    a.enable_zerocopy(8192)
    assert a.AllocateSendBuffer(0) is None and a.AllocateSendBuffer(8193) is None
    p = a.AllocateSendBuffer(3000)
    assert p is not None and a.AllocateSendBuffer(16) is None
    data = (bytes(range(256)) * 12)[:3000]
    zc_write(g, a, p, data)
    assert a.SendZerocopy([(p, 3000)]) == 3000          # Send would stop at W(staging = 2048) = 2024
    st = a.zerocopy_state()
    assert st["tail"] == 0 and st["zerocopy_bytes"] == 3000 and st["sges"] == 3
    assert a.AllocateSendBuffer(16) is not None
    assert b.Recv(4096) == data
    # host slices beside the serialised message are what the endpoint holds: accepted with a host-writable buffer
    # (sent as Send sends them), refused when the buffer is device memory no host slice can name
    assert a.SendZerocopy([b"host bytes"]) == 10
    assert b.Recv(4096) == b"host bytes"
    c, d = mk_link(g, R, 30)
    c.enable_zerocopy(8192, mem=2)
    with pytest.raises(Exception, match="host-writable"):
        c.SendZerocopy([b"host bytes"])
    a.close(); b.close(); c.close(); d.close()


@pytest.mark.parametrize("case", [(1 << 22, 30, 1 << 20), (1 << 16, 4, 3000), (4096, 30, 2500)], ids=["r4m_1mib", "r64k_3000", "r4k_2500"])
def test_the_host_serialises_into_the_buffer_and_the_device_sends_it_from_there(gpu, case):
    """The hook surface of SURVEY.md 8(f-3) end to end, as the reference's caller drives it: AllocateSendBuffer(n) ->
    the HOST writes the message through the returned pointer (what SerializeWithCachedSizesToArray does,
    proto_utils.h:78-84) -> the slice buffer of the write holds [frame header + message header: 14 bytes of host
    memory][the serialised message: a range of the zero-copy buffer] -> SendZerocopy.  Accepted bytes, work requests,
    zero-copy counters, ring image and state equal the oracle's orc_pair_send_zerocopy (pinned to the reference's own
    SendZerocopy); the peer reads the bytes the host wrote; the buffer is free again afterwards (tail 0) and the next
    message takes the same path.  Messages larger than the credit go out in pieces: SendZerocopy again from byte_idx."""
    g = gpu
    R, sge, n = case
    a, b = mk_link(g, R, sge)
    o = pyorc.OracleLink(R, sge)
    Z = max(2 * n, 4096)
    a.enable_zerocopy(Z)          # the default kind: host-writable
    o.enable_zerocopy(0, Z)
    assert a.zerocopy_mem() == 0
    rng = random.Random(n)
    for msg in range(4):
        body = bytes(rng.getrandbits(8) for _ in range(251)) * (n // 251) + bytes(n % 251)
        hdr = bytes([0, (n >> 8) & 255, n & 255, 0, 0, 0, 0, 0, 2 * msg + 1, 0]) + n.to_bytes(4, "big")
        p, off = a.AllocateSendBuffer(n), o.allocate_send_buffer(0, n)
        assert p is not None and off == 0
        C.memmove(p, body, n)                         # <- the host serialises here
        o.zerocopy_write(0, off, body)
        dsl, osl = [hdr, (p, n)], [hdr, ("zc", off, n)]
        idx, byte, got = 0, 0, b""
        while idx < 2:
            sent = a.SendZerocopy(dsl[idx:], byte)
            assert sent == o.send_zerocopy(0, osl[idx:], byte), (msg, idx, byte)
            assert a.last_wrs() == o.last_wrs(0)
            assert a.zerocopy_state() == o.zerocopy_state(0)
            assert _ring_eq(b.ring_mem(), o.ring_mem(1)), msg
            check_state(a, b, o)
            left = sent
            while left > 0:
                room = (len(hdr) if idx == 0 else n) - byte
                if left >= room:
                    left -= room
                    idx += 1
                    byte = 0
                else:
                    byte += left
                    left = 0
            while True:                               # the peer reads what has arrived (and returns credit)
                r = b.Recv(R)
                assert r == o.recv(1, R)
                if not r:
                    break
                got += r
        assert got == hdr + body
        assert a.zerocopy_state()["tail"] == 0        # the buffer is free for the next message
        assert _ring_eq(b.ring_mem(), o.ring_mem(1))
        check_state(a, b, o)
    a.close(); b.close(); o.close()


def test_zero_copy_golden_traces_on_gpu(gpu):
    """The committed reference-generated traces (tests/golden/zc_*.json) replayed on the device: allocator
    answers, accepted bytes, work requests, entry counts, buffer tail, state, ring image (padding masked: the
    device writes zeros there)."""
    import glob
    import json
    from tests.test_golden_ring import payload
    g = gpu
    here = os.path.dirname(os.path.abspath(__file__))
    for path in sorted(glob.glob(os.path.join(here, "golden", "zc_*.json"))):
        doc = json.load(open(path))
        R, Z = doc["ring_size"], doc["zerocopy_buffer"]
        a, b = mk_link(g, R, doc["max_sge"])
        o = pyorc.OracleLink(R, doc["max_sge"])     # (for the ring image: the trace stores its hash only)
        a.enable_zerocopy(Z)
        o.enable_zerocopy(0, Z)
        zc_base = None
        for i, st in enumerate(doc["steps"]):
            if st["op"] == "zc_send":
                dsl, osl, keep, allocs = [], [], [], []
                for item in st["slices"]:
                    if item[0] == "zc":
                        _, seed, n = item
                        dptr, off = a.AllocateSendBuffer(n), o.allocate_send_buffer(0, n)
                        allocs.append(off)
                        assert (dptr is None) == (off is None)
                        if dptr is not None:
                            zc_base = zc_base or dptr
                            assert dptr - zc_base == off
                        else:
                            off = (seed * 131) % (Z - n + 1)
                        if zc_base is None:  # no allocation has succeeded yet: learn the base from one
                            pytest.skip("trace starts with a refused allocation")
                        data = payload(seed, n)
                        zc_write(g, a, zc_base + off, data)
                        o.zerocopy_write(0, off, data)
                        dsl.append((zc_base + off, n))
                        osl.append(("zc", off, n))
                    else:
                        data = payload(item[0], item[1])
                        buf = g.DeviceBuffer(data=data)
                        keep.append(buf)
                        dsl.append(buf)
                        osl.append(data)
                assert allocs == st["allocs"]
                bi = st.get("byte_idx", 0)
                assert a.SendZerocopy(dsl, bi) == st["sent"] == o.send_zerocopy(0, osl, bi), (doc["name"], i)
                assert [list(w) for w in a.last_wrs()] == st["wrs"]
                assert a.zerocopy_state() == st["zc_state"]
            elif st["op"] == "send":
                data = [payload(s, n) for s, n in st["slices"]]
                assert a.Send([g.DeviceBuffer(data=d) for d in data]) == st["sent"] == o.send(0, data)
            else:
                got = b.Recv(st["cap"])
                assert got == o.recv(1, st["cap"]) and len(got) == st["got_len"]
            assert _ring_eq(b.ring_mem(), o.ring_mem(1)), (doc["name"], i)
            sa, sb = a.state(), b.state()
            assert all(sb[k] == v for k, v in st["rx_state"].items()), (doc["name"], i)
            assert all(sa[k] == v for k, v in st["tx_state"].items()), (doc["name"], i)
        a.close(); b.close(); o.close()
