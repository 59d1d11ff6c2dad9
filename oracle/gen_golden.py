#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: generates tests/golden/ring_*.json from the REFERENCE's
own ring codec (oracle/_ref/libref_ring.so = /root/reference/src/core/lib/ibverbs/ring_buffer.cc
compiled unmodified).  Run in the build container (needs /root/reference):

    python oracle/gen_golden.py

The reference tree ships no unit tests or vectors for this codec
(test/core/ibverbs/ is absent), so these reference-generated traces are the pin:
each file is a script of operations on a loop-back pair and, after every
operation, what the reference produced (bytes accepted, work requests, ring
image as hex, reader state, delivered bytes)."""
import hashlib
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyorc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def payload(seed, n):
    rng = random.Random(seed)
    return bytes(rng.getrandbits(8) for _ in range(n))


def trace(name, ring, max_sge, ops):
    link = pyorc.RefLink(ring, max_sge)
    steps = []
    for op in ops:
        rec = dict(op)
        if op["op"] == "send":
            slices = [payload(s, n) for s, n in op["slices"]]
            rec["sent"] = link.send(0, slices, op.get("byte_idx", 0))
            rec["wrs"] = link.last_wrs(0)
            rec["staging_sha256"] = hashlib.sha256(link.staging_mem(0)).hexdigest()
        elif op["op"] == "recv":
            got = link.recv(1, op["cap"])
            rec["got_len"] = len(got)
            rec["got_sha256"] = hashlib.sha256(got).hexdigest()
        elif op["op"] == "endpoint_read":
            got, alloc = link.endpoint_read(1)
            rec["got_len"] = len(got)
            rec["alloc"] = alloc
            rec["got_sha256"] = hashlib.sha256(got).hexdigest()
        ring_img = link.ring_mem(1)
        rec["ring_sha256"] = hashlib.sha256(ring_img).hexdigest()
        if ring <= 256:
            rec["ring_hex"] = ring_img.hex()
        rec["rx_state"] = link.state(1)
        rec["tx_state"] = link.state(0)
        rec["readable"] = link.readable(1)
        rec["has_message"] = link.has_message(1)
        rec["writable"] = link.writable(0)
        steps.append(rec)
    link.close()
    doc = {"name": name, "ring_size": ring, "max_sge": max_sge,
           "generator": "oracle/gen_golden.py over oracle/_ref/libref_ring.so "
                        "(reference src/core/lib/ibverbs/ring_buffer.cc)",
           "payload_rule": "random.Random(seed).getrandbits(8) per byte, slices = [seed, length]",
           "steps": steps}
    with open(os.path.join(OUT, "ring_%s.json" % name), "w") as f:
        json.dump(doc, f, indent=0, separators=(",", ":"))
    return doc


def zc_slices(link, zc_cap, spec):
    """spec: [["zc", seed, n] | [seed, n]] -> slices for send_zerocopy.  A "zc" slice asks the allocator
    first (AllocateSendBuffer); when that refuses (the buffer is not empty) the bytes go to the range at
    (seed * 131) % (zc_cap - n + 1), which SendZerocopy still treats as inside the buffer."""
    out, allocs = [], []
    for item in spec:
        if item[0] == "zc":
            _, seed, n = item
            off = link.allocate_send_buffer(0, n)
            allocs.append(off)
            if off is None:
                off = (seed * 131) % (zc_cap - n + 1)
            link.zerocopy_write(0, off, payload(seed, n))
            out.append(("zc", off, n))
        else:
            out.append(payload(item[0], item[1]))
    return out, allocs


def trace_zerocopy(name, ring, max_sge, zc_cap, ops):
    """Same as trace() with PairPollable::AllocateSendBuffer / SendZerocopy (pair.cc:305-323, 793-941) as
    transcribed in oracle/ref_driver.cc over the reference-built ring codec.  Files are named zc_*.json (the
    ring_*.json replays do not know these operations)."""
    link = pyorc.RefLink(ring, max_sge)
    link.enable_zerocopy(0, zc_cap)
    steps = []
    for op in ops:
        rec = dict(op)
        if op["op"] == "zc_send":
            slices, allocs = zc_slices(link, zc_cap, op["slices"])
            rec["allocs"] = allocs
            rec["sent"] = link.send_zerocopy(0, slices, op.get("byte_idx", 0))
            rec["wrs"] = link.last_wrs(0)
            rec["zc_state"] = link.zerocopy_state(0)
            rec["staging_sha256"] = hashlib.sha256(link.staging_mem(0)).hexdigest()
        elif op["op"] == "send":
            rec["sent"] = link.send(0, [payload(s, n) for s, n in op["slices"]], 0)
        elif op["op"] == "recv":
            got = link.recv(1, op["cap"])
            rec["got_len"] = len(got)
            rec["got_sha256"] = hashlib.sha256(got).hexdigest()
        ring_img = link.ring_mem(1)
        rec["ring_sha256"] = hashlib.sha256(ring_img).hexdigest()
        if ring <= 256:
            rec["ring_hex"] = ring_img.hex()
        rec["rx_state"] = link.state(1)
        rec["tx_state"] = link.state(0)
        steps.append(rec)
    link.close()
    doc = {"name": name, "ring_size": ring, "max_sge": max_sge, "zerocopy_buffer": zc_cap,
           "generator": "oracle/gen_golden.py: PairPollable::SendZerocopy as transcribed in oracle/ref_driver.cc "
                        "over oracle/_ref/libref_ring.so (reference src/core/lib/ibverbs/ring_buffer.cc)",
           "payload_rule": "random.Random(seed).getrandbits(8) per byte; slices = [seed, length] or "
                           "['zc', seed, length] (see zc_slices in oracle/gen_golden.py)",
           "steps": steps}
    with open(os.path.join(OUT, "zc_%s.json" % name), "w") as f:
        json.dump(doc, f, indent=0, separators=(",", ":"))
    return doc


def main():
    os.makedirs(OUT, exist_ok=True)
    # 1. the probe of SURVEY.md section 8c: "hello ring\0" -> tail 32, readable 11
    trace("hello", 128, 30, [
        {"op": "send", "slices": [[1, 11]]},
        {"op": "recv", "cap": 11},
    ])
    # 2. tiny ring: every wrap position, credit every 32 bytes
    ops = []
    rng = random.Random(42)
    for i in range(60):
        r = rng.random()
        if r < 0.5:
            ops.append({"op": "send", "slices": [[1000 + i * 10 + k, rng.choice([1, 2, 7, 8, 9, 15, 16, 17, 24, 40])]
                                                 for k in range(rng.randint(1, 4))],
                        "byte_idx": 0})
        elif r < 0.8:
            ops.append({"op": "recv", "cap": rng.choice([1, 3, 8, 64])})
        else:
            ops.append({"op": "endpoint_read"})
    trace("tiny_wrap", 64, 3, ops)
    # 3. max_sge = 30 batching + byte_idx + partial writes on a 4 KiB ring
    ops = []
    for i in range(40):
        r = rng.random()
        if r < 0.55:
            sl = [[2000 + i * 50 + k, rng.choice([5, 9, 14, 100, 256, 257, 1000, 1365, 4096])]
                  for k in range(rng.randint(1, 40))]
            ops.append({"op": "send", "slices": sl,
                        "byte_idx": rng.randrange(sl[0][1]) if rng.random() < 0.3 else 0})
        elif r < 0.75:
            ops.append({"op": "recv", "cap": rng.choice([1, 100, 256, 4096])})
        else:
            ops.append({"op": "endpoint_read"})
    trace("sge30_4k", 4096, 30, ops)
    # 4. HTTP/2-shaped slices (9-byte frame header slice + 16 KiB payload slice) on 1 MiB
    ops = []
    for i in range(6):
        sl = [[3000 + i, 14], [3100 + i, 16379]]
        for f in range(20):
            sl += [[3200 + i * 100 + f, 9], [3300 + i * 100 + f, 16384]]
        sl += [[3400 + i, 9], [3500 + i, 9]]
        ops.append({"op": "send", "slices": sl})
        for _ in range(50):
            ops.append({"op": "endpoint_read"})
    trace("h2_shaped_1m", 1 << 20, 30, ops)
    # 5. zero-copy send buffer: small ring with every wrap position, and a 4 KiB ring with max_sge 5
    for name, ring, sge, zc_cap, sizes, n_ops in (("tiny", 128, 8, 96, [1, 2, 7, 8, 9, 15, 16, 17, 24, 40, 90], 70),
                                                  ("sge5_4k", 4096, 5, 8192, [5, 9, 100, 256, 257, 1000, 1365, 3000, 4096], 50)):
        zrng = random.Random(77)
        ops = []
        for i in range(n_ops):
            r = zrng.random()
            if r < 0.55:
                sl = []
                for k in range(zrng.randint(1, 4)):
                    n = zrng.choice(sizes)
                    seed = 5000 + i * 10 + k
                    sl.append(["zc", seed, min(n, zc_cap)] if zrng.random() < 0.6 else [seed, n])
                first = sl[0][2] if sl[0][0] == "zc" else sl[0][1]
                ops.append({"op": "zc_send", "slices": sl,
                            "byte_idx": zrng.randrange(first) if zrng.random() < 0.3 else 0})
            elif r < 0.65:
                ops.append({"op": "send", "slices": [[6000 + i, zrng.choice(sizes)]]})
            else:
                ops.append({"op": "recv", "cap": zrng.choice([1, 8, 64, 256, ring])})
        trace_zerocopy(name, ring, sge, zc_cap, ops)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
