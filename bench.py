#!/usr/bin/env python3
"""bench.py -- streaming GiB/s @ 1 MiB messages through the HIP ring-buffer endpoint.

One "step" = one pass of the hot path over one batch: `--msgs` gRPC messages of
1 MiB payload, framed into HTTP/2 DATA frames exactly as chttp2 hands them to
grpc_endpoint_write (130 slices per message), pushed through ONE connection:
slice gather + ring-record encode -> wire -> message-ready detection -> record
decode + scatter + zero-fill, with the head/tail credit protocol running between
the two ends.  Inputs are resident in HBM before the timed region; the delivered
slices stay in HBM (the PCIe-inclusive rate is reported separately in DESIGN.md).

Contract: python bench.py --gpus N --steps K --warmup W  (N>1 under torch.distributed.run,
one rank per GPU, one independent connection per GPU: weak scaling, no collective
on the data path).  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MIB = 1 << 20
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def proto_message(i, payload=MIB, prng_seed=None):
    """Serialized SimpleRequest{bytes message = <payload bytes>} (micro_benchmark.proto):
    tag 0x0a, varint length, bytes.  Content rotates with the message index so a
    misordered or stale delivery cannot compare equal; with prng_seed the body is PRNG bytes
    (the reference's second payload kind, examples/cpp/micro-bench: seed 1234)."""
    n = payload
    var = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        var.append(b | (0x80 if n else 0))
        if not n:
            break
    if prng_seed is not None:
        import random
        body = random.Random(prng_seed + i).randbytes(payload)
    else:
        body = bytes(((j + 17 * i) % 251) for j in range(256)) * (payload // 256 + 1)
    return bytes([0x0A]) + bytes(var) + body[:payload]


class Workload:
    """B framed messages laid out in HBM + the slice list of the endpoint_write."""

    def __init__(self, g, n_msgs, payload=MIB, max_frame=16384, stream_id=1, prng_seed=None):
        self.payload = payload
        from grpc_rdma_amd import h2
        self.g = g
        self.n_msgs = n_msgs
        self.msgs = [proto_message(i, payload, prng_seed) for i in range(min(n_msgs, 8))]
        self.msg_len = len(self.msgs[0])
        layout = h2.frame_message(self.msg_len, stream_id, max_frame)
        self.layout = layout
        self.slices_per_msg = len(layout)
        # HBM: one buffer per distinct message content, n_msgs copies of the payload
        self.payload_buf = g.DeviceBuffer(nbytes=n_msgs * self.msg_len)
        lib = g.load()
        for i in range(n_msgs):
            src = self.msgs[i % len(self.msgs)]
            b = C.create_string_buffer(src, len(src))
            lib.grdma_copy_to_device(self.payload_buf.ptr + i * self.msg_len, b, len(src))
        # inlined slices: bytes live at offset 9 of a 32-byte grpc_slice (slice.h:60-75)
        n_inl = sum(1 for it in layout if it[0] == "inl")
        hdr_host = bytearray(32 * n_inl * n_msgs)
        self.slices = []
        self.wire_per_msg = []
        k = 0
        for i in range(n_msgs):
            for it in layout:
                if it[0] == "inl":
                    off = 32 * k + 9
                    hdr_host[off:off + len(it[1])] = it[1]
                    self.slices.append(("h", off, len(it[1])))
                    k += 1
                else:
                    self.slices.append(("p", i * self.msg_len + it[1][0], it[1][1]))
        self.hdr_buf = g.DeviceBuffer(data=bytes(hdr_host))
        self.sge = [((self.hdr_buf.ptr if kind == "h" else self.payload_buf.ptr) + off, n)
                    for kind, off, n in self.slices]
        self.lens = [n for _, _, n in self.slices]
        self.N = sum(self.lens)                       # endpoint payload bytes per step
        self.E = h2.ring_bytes_for(self.lens)         # encoded ring bytes per step
        self.user_bytes = n_msgs * payload            # application payload per step

    def expected_wire(self, i):
        """Byte stream of message i as it must come out of the endpoint."""
        m = self.msgs[i % len(self.msgs)]
        return b"".join(it[1] if it[0] == "inl" else m[it[1][0]:it[1][0] + it[1][1]]
                        for it in self.layout)


class MixedWorkload(Workload):
    """Message sizes drawn like the reference's random-size test (examples/cpp/test/common.h:4-31:
    uniform in [1, 4 MiB - 1 KiB]), fixed seed; every message framed on its own stream id."""

    def __init__(self, g, n_msgs, seed=0):
        import random
        from grpc_rdma_amd import h2
        rng = random.Random(seed)
        self.sizes = [rng.randint(1, (4 << 20) - 1024) for _ in range(n_msgs)]
        self.g, self.n_msgs, self.payload = g, n_msgs, None
        total = sum(self.sizes)
        self.payload_buf = g.DeviceBuffer(nbytes=total + 64)
        self.block = bytes((i * 7 + 1) % 251 for i in range(1 << 20))
        lib = g.load()
        off = 0
        for n in self.sizes:
            for o in range(0, n, len(self.block)):
                k = min(len(self.block), n - o)
                lib.grdma_copy_to_device(self.payload_buf.ptr + off + o, self.block, k)
            off += n
        hdr, self.slices, k, base = bytearray(), [], 0, 0
        self.layouts = []
        for i, n in enumerate(self.sizes):
            lay = h2.frame_message(n, 2 * i + 1, 16384)
            self.layouts.append(lay)
            for it in lay:
                if it[0] == "inl":
                    o = 32 * k + 9
                    hdr += bytes(32)
                    hdr[o:o + len(it[1])] = it[1]
                    self.slices.append(("h", o, len(it[1])))
                    k += 1
                else:
                    self.slices.append(("p", base + it[1][0], it[1][1]))
            base += n
        self.hdr_buf = g.DeviceBuffer(data=bytes(hdr) + bytes(64))
        self.sge = [((self.hdr_buf.ptr if kind == "h" else self.payload_buf.ptr) + o, n) for kind, o, n in self.slices]
        self.lens = [n for _, _, n in self.slices]
        self.N = sum(self.lens)
        self.E = h2.ring_bytes_for(self.lens)
        self.user_bytes = total
        self.slices_per_msg = len(self.lens) // n_msgs

    def expected_wire(self, i):
        n = self.sizes[i]
        # every message restarts the block pattern at each 1 MiB boundary of its own payload
        m = b"".join(self.block[:min(len(self.block), n - o)] for o in range(0, n, len(self.block)))
        return b"".join(it[1] if it[0] == "inl" else m[it[1][0]:it[1][0] + it[1][1]] for it in self.layouts[i])


def run_json(cmd, timeout, env=None):
    """Run a helper process that prints one JSON line; None (with the reason) if it cannot."""
    import subprocess
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        if r.returncode != 0:
            return {"error": ("rc %d: " % r.returncode) + (r.stderr or r.stdout)[-200:]}
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # missing binary, timeout, no loop-back networking ...
        return {"error": str(e)[:200]}


def tcp_baseline():
    """What the reference's TCP platform can reach on these host cores (BASELINE.md quotes the
    RDMA modes relative to TCP): a real gRPC stack over loop-back TCP (the grpcio wheel: gRPC C-core,
    Python binding on top) and the raw sendmsg/recvmsg floor under any TCP transport
    (oracle/tcp_floor.c), in the two shapes of the metric."""
    floor = os.path.join(ROOT, "oracle", "_build", "tcp_floor")
    gl = os.path.join(ROOT, "oracle", "grpcio_loopback.py")
    fs = run_json([floor, "stream", "4000", str(MIB)], 60)
    fp = run_json([floor, "pingpong", "100000", "80"], 60)
    gs_ = run_json([sys.executable, gl, "stream", "4", str(MIB)], 90)
    gu = run_json([sys.executable, gl, "unary", "4", "66"], 90)
    return {"nproc": os.cpu_count(),
            "sendmsg_floor": {"stream_GiBps": fs.get("GiBps"), "rtt_p50_us": fp.get("p50_us"), "cores": 2,
                              "what": "raw sendmsg (130 iovecs per 1 MiB message) / recvmsg over loop-back TCP, "
                                      "writer + reader thread; 80-byte ping-pong, 100k round trips",
                              "error": fs.get("error") or fp.get("error")},
            "grpcio_loopback": {"stream_GiBps": gs_.get("GiBps"), "unary_p50_us": gu.get("p50_us"),
                                "cores": "client thread + C-core poller + 2 server workers (not pinned)",
                                "what": "grpcio %s client-streaming 1 MiB messages for 4 s / unary 66-byte echo for 4 s, "
                                        "one insecure channel on 127.0.0.1, identity serializers" % gs_.get("grpcio"),
                                "error": gs_.get("error") or gu.get("error")}}


def conn_setup_us(g, ring=4 << 20, max_sge=30, n=8):
    """Connection set-up + tear-down (grdma_pair_create + _destroy of the reference's default shape): every block from
    the allocator vs from the PairPool (pair.h:273-333; here the pool keeps the memory of closed connections)."""
    lib = g.load()
    lib.grdma_pair_pool_trim()

    def lap(k):
        t0 = time.perf_counter()
        for _ in range(k):
            p = lib.grdma_pair_create(ring, max_sge, 0)
            if not p:
                raise RuntimeError(lib.grdma_last_error().decode())
            lib.grdma_pair_destroy(p)
        return 1e6 * (time.perf_counter() - t0) / k

    lap(2)
    cold = lap(n)
    lib.grdma_pair_pool_reserve(1, ring, max_sge, 0, 0)
    lap(2)
    pooled = lap(n)
    lib.grdma_pair_pool_trim()
    return {"ring_kib": ring >> 10, "allocator": round(cold, 1), "pair_pool": round(pooled, 1)}


def err_text(e):
    """Exception -> short text that still says where it happened (an assert carries no message)."""
    import traceback
    tb = traceback.extract_tb(e.__traceback__)
    where = "%s:%d" % (os.path.basename(tb[-1].filename), tb[-1].lineno) if tb else "?"
    return ("%s at %s: %s" % (type(e).__name__, where, e))[:240]


def measured_copy_ceiling(torch, dev, nbytes=256 * MIB, reps=20):
    """What this GPU's own copy engines / fill kernels reach on a buffer of the step's size:
    the practical ceiling to read roofline.frac against (the 8 TB/s peak is the spec figure)."""
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    b = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    a.zero_(); b.zero_()
    out = {}
    for name, fn, traffic in (("hipMemcpyDtoD", lambda: b.copy_(a), 2 * nbytes), ("hipMemset", lambda: b.zero_(), nbytes)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        sec = e0.elapsed_time(e1) * 1e-3 / reps
        out[name + "_traffic_GBps"] = round(traffic / sec / 1e9, 1)
    out["bytes"] = nbytes
    return out


def cpu_baseline(wl, ring, max_sge, target_s=12.0, threads=1):
    """The CPU port (oracle/) timed on host cores: same slices, same ring size, full
    pair protocol + endpoint read loop, single thread (the reference is
    single-reader/single-writer per pair).  threads > 1: that many connections, each on a thread of its own (the C
    loop runs outside the interpreter lock), messages counted over the wall time of all of them -- how the reference's
    codec scales over the host's cores when the connections are independent."""
    from oracle import pyorc
    wire = wl.expected_wire(0)
    lens = wl.lens[:wl.slices_per_msg]
    # the reference's own ring codec (oracle/_ref, built from /root/reference where that exists
    # and shipped as a prebuilt file) when it is there, the plain-C port otherwise
    kind, run, what = "port", pyorc.stream_baseline, "oracle/ pair + endpoint-read loop"
    if pyorc.ref_available():
        kind, what = "reference", "the reference-built ring codec (oracle/_ref) in the pair + endpoint-read loop"
        run = lambda *a: pyorc.ref_stream_baseline(*a)[:2]
    n, sec = run(ring, max_sge, wire, lens, 16)
    per_msg = sec / 16
    n_msgs = max(16, min(400000, int(target_s / per_msg)))
    if threads > 1:
        import threading
        secs = [0.0] * threads

        def one(k):
            secs[k] = run(ring, max_sge, wire, lens, n_msgs)[1]
        ts = [threading.Thread(target=one, args=(k,)) for k in range(threads)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        sec = time.perf_counter() - t0
        gib = (threads * n_msgs * (wl.user_bytes // wl.n_msgs)) / sec / (1 << 30)
        return {"value": round(gib, 3), "unit": "GiB/s", "cores": threads, "kind": kind,
                "sample": "%d connections x %d x 1 MiB messages (%d slices each) through %s, %d KiB rings, one thread per "
                          "connection, %.1f s wall (slowest thread %.1f s)" % (threads, n_msgs, len(lens), what, ring >> 10, sec, max(secs))}
    n, sec = run(ring, max_sge, wire, lens, n_msgs)
    gib = (n_msgs * (wl.user_bytes // wl.n_msgs)) / sec / (1 << 30)
    return {"value": round(gib, 3), "unit": "GiB/s", "cores": 1, "kind": kind,
            "sample": "%d x 1 MiB messages (%d slices each) through %s, %d KiB ring, 1 thread, %.1f s" % (
                n_msgs, len(lens), what, ring >> 10, sec)}


def sources_sha16():
    """sha256 (16 hex digits) over the product's kernel and host sources: what ties a committed counter summary
    (profiles/r0N_pmc_*.json, tools/pmc_summary.py) to the library a bench run is timing."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "grpc-rdma_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "include", "*")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_summary_for(ring_kb):
    """The newest committed counter summary for this ring size and whether it was collected from the sources this
    run is timing: -> (path or None, summary dict or None, stale: str or None)."""
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        p = os.path.join(ROOT, "profiles", "%s_pmc_ring%dm_summary.json" % (rnd, ring_kb // 1024))
        if os.path.exists(p):
            try:
                d = json.load(open(p))
            except Exception:
                return p, None, "unreadable"
            have, want = d.get("sources_sha16"), sources_sha16()
            if have != want:
                return p, None, "collected from other sources (%s, this run %s): not used" % (have, want)
            return p, d, None
    return None, None, "no counter summary committed"


def measure_rtt(g, iters=100000, warmup=2000, reads="standing"):
    """Unary ping-pong, 64-byte payload both ways, 1 connection, 4 MiB rings in HBM, host
    in the loop where gRPC's consumer is (host slices in, host-visible slices out).  A write is one command to the
    resident latency engine; a read is what gRPC keeps outstanding on every connection -- a STANDING order
    (grdma_pair_arm_read) that a watcher workgroup of the engine carries out when the bytes land in the pair's own ring
    (k_watch: arrival-triggered, whoever wrote them), so that the host's read is a look at pinned memory.  Every record
    goes through the ring.  reads="command": round 4's way, every read a command of its own (four commands per round
    trip).  p50/p95/p99 from the sorted samples."""
    from grpc_rdma_amd import h2
    import ctypes as C
    lib = g.load()
    msg = bytes([0x0A, 64]) + bytes(range(64))           # SimpleRequest{bytes message = 64 B}
    items = h2.frame_message(len(msg), 1)
    slices = [i[1] if i[0] == "inl" else msg[i[1][0]:i[1][0] + i[1][1]] for i in items]
    a, b = g.Pair(4 << 20, 30), g.Pair(4 << 20, 30)
    g.connect_pairs(a, b)
    a.set_latency_mode(True)
    b.set_latency_mode(True)
    if reads == "standing":
        a.arm_read(64)
        b.arm_read(64)
    g._lib.check(lib.grdma_engine_start())
    lib.grdma_watch_fast_drains.restype = C.c_uint64
    fd0 = int(lib.grdma_watch_fast_drains())
    prof = {}
    try:
        t0 = time.perf_counter()
        rtt, ph = g.pingpong(a, b, slices, slices, iters=iters, warmup=warmup)
        wall = time.perf_counter() - t0
        counts = {"watch_hits": a.watch_hits() + b.watch_hits(),
                  "single_wave_drains": int(lib.grdma_watch_fast_drains()) - fd0,
                  "watcher_workgroups": int(lib.grdma_engine_watchers())}
        # a second, short pass with the GRPCProfiler mirror on (slot 0): the same round trips under the
        # reference's op names (include/grpcpp/stats_time.h:11-44)
        try:
            import ctypes as C
            lib.grdma_stats_time_get.restype = C.c_uint64
            lib.grdma_stats_time_get.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double)]
            lib.grdma_stats_time_op_name.restype = C.c_char_p
            lib.grdma_stats_time_op_name.argtypes = [C.c_int]
            lib.grdma_stats_time_init.argtypes = [C.c_int]
            lib.grdma_stats_time_init(0)
            lib.grdma_stats_time_enable()
            g.pingpong(a, b, slices, slices, iters=max(100, iters // 10), warmup=10)
            lib.grdma_stats_time_disable()
            for op in range(31):
                st = (C.c_double * 5)()
                n = lib.grdma_stats_time_get(0, op, st)
                if n:
                    prof[lib.grdma_stats_time_op_name(op).decode()] = {
                        "count": int(n), "mean": round(st[0] / 1e3, 2), "p50": round(st[1] / 1e3, 2),
                        "p95": round(st[2] / 1e3, 2), "p99": round(st[3] / 1e3, 2)}
            lib.grdma_stats_time_shutdown()
        except Exception as e:  # the breakdown is an extra: never lose the round-trip numbers over it
            prof = {"error": str(e)[:160]}
    finally:
        lib.grdma_engine_stop()
    a.close()
    b.close()
    rtt.sort()
    if reads != "standing":  # (the comparison leg: round 4's configuration under its own keys)
        return {"rtt_read_commands_p50_us": round(rtt[iters // 2] / 1e3, 2),
                "rtt_read_commands_p95_us": round(rtt[int(iters * .95)] / 1e3, 2),
                "rtt_read_commands_iters": iters,
                "rtt_read_commands_breakdown_us": {k: round(v / iters / 1e3, 2) for k, v in zip(
                    ["client_write", "server_read", "server_write", "client_read"], ph)}}
    return {"rtt_p50_us": round(rtt[iters // 2] / 1e3, 2), "rtt_p95_us": round(rtt[int(iters * .95)] / 1e3, 2),
            "rtt_p99_us": round(rtt[int(iters * .99)] / 1e3, 2), "rtt_p10_us": round(rtt[iters // 10] / 1e3, 2),
            "rtt_mean_us": round(sum(rtt) / iters / 1e3, 2),
            "rtt_iters": iters, "rtt_seconds": round(wall, 2),
            "rtt_config": "unary ping-pong 64 B, 1 connection, 4 MiB ring in HBM, slices [14 B][66 B] each way; a write = "
                          "one command to the resident latency engine, a read = a standing order carried out by a watcher "
                          "workgroup when the bytes land in the pair's own ring (arrival-triggered, k_watch) -- every record "
                          "goes through the ring; host slices in / pinned slices out",
            "rtt_counts": counts,
            "rtt_breakdown_us": {k: round(v / iters / 1e3, 2) for k, v in zip(
                ["client_write", "server_read", "server_write", "client_read"], ph)},
            "rtt_profile_us": prof}


def rtt_subprocess(flag, iters, timeout_s=150, key="rtt_error"):
    """The ping-pong legs run a RESIDENT kernel (k_engine).  One that wedged would hang every later synchronize of
    this process and take the whole line with it, so each leg runs in a child that can be killed."""
    import subprocess
    try:
        cp = subprocess.run([sys.executable, os.path.abspath(__file__), flag, "--rtt-iters", str(iters)],
                            capture_output=True, text=True, timeout=timeout_s, stdin=subprocess.DEVNULL)
        for line in reversed(cp.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {key: ("exit %d: " % cp.returncode) + (cp.stderr or cp.stdout)[-300:]}
    except subprocess.TimeoutExpired:
        return {key: "timed out after %d s (child killed)" % timeout_s}
    except Exception as e:
        return {key: err_text(e)}


EMULATED = False  # (set by main: GRDMA_BENCH_EMULATED=1, the dry run of the CPU suite)


def dev_sync(torch):
    if not EMULATED:
        torch.cuda.synchronize()


def fanout_leg(g, gs, grp, torch, args, flags, n_msgs=128):
    """One 128 MiB stream (128 x 1 MiB messages) ingested on rank 0, decoded into a torch
    tensor, then rebalanced to all ranks with one grouped RCCL send/recv step over views of the arena
    (grpc_rdma_amd.fanout: no padded copies on the source)."""
    from grpc_rdma_amd import fanout
    dev = torch.device("cpu") if EMULATED else torch.device("cuda", grp.local_rank)
    ring = args.ring_kb * 1024
    arena, slices = torch.zeros(16, dtype=torch.uint8, device=dev), []
    t_ingest = 0.0
    if grp.rank == 0:
        wl = Workload(g, n_msgs)
        tx, rx = g.Pair(ring, args.max_sge, flags), g.Pair(ring, args.max_sge, flags)
        g.connect_pairs(tx, rx)
        dst_cap = wl.N + 16 * (len(wl.lens) * 2 + 64) + 4096
        arena = torch.empty(dst_cap, dtype=torch.uint8, device=dev)
        job = gs.StreamJob(tx, rx, wl.sge, arena.data_ptr(), dst_cap, len(wl.lens) * 2 + 64,
                           max(8, 4 * (wl.E // (ring // 2) + 2)))
        r = job.run(gs.RUN_EAGER)
        assert r.done
        job.set_rounds(int(max(r.tx_rounds, r.rx_rounds)))
        job.run(gs.RUN_GRAPH)
        dev_sync(torch)
        t0 = time.perf_counter()
        job.launch()
        job.sync()
        t_ingest = time.perf_counter() - t0
        slices = job.delivered_slices()
    dev_sync(torch)
    grp.barrier()
    t0 = time.perf_counter()
    mine, my_slices, start = fanout.scatter_arena(grp, arena, slices, src=0, return_start=True)
    dev_sync(torch)
    t_scatter = grp.max(time.perf_counter() - t0)
    got = grp.sum(sum(n for _, n in my_slices))
    # what the ranks hold afterwards, concatenated in rank order, must BE the framed stream: every rank checksums its
    # share where it lies (byte sum + position-weighted sum, the position counted from the start of the whole stream),
    # the partial sums are added over the ranks and compared with the same two sums over the messages as framed on the host
    c1, c2 = fanout.stream_checksum(mine, my_slices, start)
    got_c1, got_c2 = grp.sum_int(c1), grp.sum_int(c2)
    out = {}
    if grp.rank == 0:
        total = sum(n for _, n in slices)
        import numpy as np
        e1 = e2 = pos = 0
        for i in range(wl.n_msgs):
            b = np.frombuffer(wl.expected_wire(i), dtype=np.uint8).astype(np.int64)
            e1 += int(b.sum())
            e2 += int((b * ((np.arange(pos, pos + b.size, dtype=np.int64) % 65521) + 1)).sum())
            pos += b.size
        out = {"fanout_config": "%d MiB stream ingested on GPU 0, one grouped RCCL send/recv step to %d GPUs" % (n_msgs, grp.world),
               "fanout_ingest_ms": round(1e3 * t_ingest, 3), "fanout_scatter_ms": round(1e3 * t_scatter, 3),
               "fanout_GiBps": round(n_msgs * MIB / (t_ingest + t_scatter) / (1 << 30), 3),
               "fanout_bytes_ok": bool(got == total),
               "fanout_checksum_ok": bool(got == total == pos and (got_c1, got_c2) == (e1, e2)),
               "fanout_checksum": {"byte_sum": got_c1, "position_weighted_sum": got_c2, "expected": [e1, e2],
                                   "what": "sum over the ranks of each rank's checksum of its share (grpc_rdma_amd.fanout."
                                           "stream_checksum) against the framed messages summed on the host"}}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--msgs", type=int, default=1008,
                    help="1 MiB messages per step of the headline leg: 1008 = sixteen FULL rounds of two Sends of <= 4095 slices "
                         "(63 messages each) -- a continuous stream has no short last round and no per-step first gather / last "
                         "scatter; rounds 1-5 timed steps of 256 messages (four rounds of 63 and one of 4), which stays in "
                         "the line as value_msgs256_per_step")
    ap.add_argument("--leg-msgs", type=int, default=256, help="1 MiB messages per step of the comparison legs")
    ap.add_argument("--ring-kb", type=int,
                    default=int(os.environ.get("GRPC_RDMA_RING_BUFFER_SIZE_KB", 262144)),
                    help="ring size (GRPC_RDMA_RING_BUFFER_SIZE_KB); the reference default is 4096")
    ap.add_argument("--max-sge", type=int, default=4095)
    ap.add_argument("--sends", type=int, default=2,
                    help="consecutive Sends per round of the pipelined single-connection legs (rdma_flush sends again while "
                         "the ring has room; grdma_stream_job_set_sends); 1 = one Send per round")
    ap.add_argument("--promise", action="store_true",
                    help="the headline leg with the promised credit (grdma_stream_job_set_promised_credit): what the "
                         "value_ring4096_sge30 leg runs with --ring-kb 4096 --max-sge 30 --sends 64 (profiling)")
    ap.add_argument("--launch", choices=["graph", "streams"], default="graph",
                    help="replay a step as one HIP graph, or issue its kernels on the job's streams")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="1: the rounds of a step overlap (plan/gather of round t+1 and scatter of "
                         "round t beside the wire + ring walk); 0: five kernels per round in order")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the sequential-schedule and direct-wire comparison runs")
    ap.add_argument("--wire", choices=["staged", "direct"], default="staged")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-small-ring", action="store_true", help="skip the extra 4 MiB-ring run")
    ap.add_argument("--no-rtt", action="store_true", help="skip the 64 B ping-pong leg")
    ap.add_argument("--no-fanout", action="store_true", help="skip the RCCL fan-out leg (N>1 only)")
    ap.add_argument("--conns", type=int, default=32,
                    help="connections per GPU in the multi-connection leg (BASELINE configs[3]: 32); 1 = skip")
    ap.add_argument("--schedule", choices=["graph"], default="graph",
                    help="how the timed step is launched: the round-by-round kernels of a HIP graph (the persistent link "
                         "engine, k_link -- a third of this rate in rounds 2 - 4 -- was retired in round 5)")
    ap.add_argument("--no-tcp-baseline", action="store_true", help="skip the loop-back TCP baselines")
    ap.add_argument("--reps", type=int, default=3,
                    help="timed regions of --steps steps each; the median is reported (reference protocol: 3 repetitions)")
    ap.add_argument("--rtt-iters", type=int, default=200000,
                    help="64 B round trips (>= 10 s of them on this part; the reference runs >= 10 s or 1 M RPCs behind "
                         "10 000 warm-up calls, examples/cpp/micro-bench/mb_client.cc:41-44)")
    ap.add_argument("--payload", type=int, default=MIB,
                    help="payload bytes of a message of the headline leg (the metric is quoted at 1 MiB; smaller values are "
                         "for the emulated dry run of the CPU suite, tests/test_bench_emulated.py)")
    ap.add_argument("--conn-msgs", type=int, default=0, help="messages per connection in the multi-connection legs (0 = max(8, 2048 / conns))")
    ap.add_argument("--fanout-msgs", type=int, default=128, help="1 MiB messages of the single-stream fan-out leg")
    ap.add_argument("--rtt-commands-only", action="store_true", help="(internal) run only the ping-pong whose reads are commands")
    ap.add_argument("--rtt-only", action="store_true", help="(internal) run only the 64 B ping-pong leg")
    ap.add_argument("--h2-only", action="store_true", help="(internal) run only the with-h2 legs")
    ap.add_argument("--h2-bulk-pairs-only", action="store_true", help="(internal) run only the 32-frames-per-bulk-step h2 leg")
    args = ap.parse_args()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # Multi-GPU runs measure the headline (and the fan-out leg) only: the comparison legs are
        # single-GPU questions, and a leg that failed on one rank alone would leave the others waiting
        # in its barrier.
        args.no_extra_legs = True
        args.no_small_ring = True
        # (the BASELINE configs[3] leg stays: --conns connections per rank -- 32 x 8 GPUs = 256 -- of 64 KiB messages,
        # every rank its own share, no data-path collective; every rank runs it, its barriers are the contract's)

    if args.rtt_only or args.rtt_commands_only:  # a child of rtt_subprocess: no torch, one leg, one JSON line
        import __graft_entry__ as ge
        import grpc_rdma_amd as g
        g.init(int(os.environ.get("LOCAL_RANK", "0")))
        if not os.environ.get("GRDMA_NO_NUMA_PIN"):  # like `numactl --cpunodebind` on the GPU's NUMA node
            g.load().grdma_host_pin_to_device_node()
        if args.rtt_only:
            print(json.dumps(measure_rtt(g, iters=args.rtt_iters, warmup=min(10000, max(10, args.rtt_iters // 10)))))
        else:
            print(json.dumps(measure_rtt(g, iters=args.rtt_iters, warmup=min(1000, max(10, args.rtt_iters // 10)), reads="command")))
        return

    import torch
    from_env = (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
                int(os.environ.get("WORLD_SIZE", "1")))
    rank, local_rank, world = from_env
    # GRDMA_BENCH_EMULATED=1 (tests/test_bench_emulated.py, the CPU suite): a DRY RUN of this script, rank for rank as the
    # driver launches it, with the product sources compiled over the wave emulator as the library (GRDMA_LIB_PATH must
    # name it), CPU tensors and gloo in place of HBM tensors and RCCL.  It checks that the N > 1 path executes -- sharded
    # connections, barriers, the max over ranks, the fan-out with its checksum -- not how fast anything is: the line it
    # prints says "emulated" in `data` and its `value` is not a measurement.  Never set on a GPU box.
    global EMULATED
    EMULATED = os.environ.get("GRDMA_BENCH_EMULATED") == "1"
    if EMULATED and "emu" not in os.path.basename(os.environ.get("GRDMA_LIB_PATH", "")):
        raise SystemExit("GRDMA_BENCH_EMULATED=1 needs GRDMA_LIB_PATH=<the emulated library>: refusing to dry-run the product library")
    import __graft_entry__ as ge
    if EMULATED:
        device = torch.device("cpu")
        local_rank_dev = 0
    else:
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        local_rank_dev = local_rank
        if not os.path.exists(ge.LIB):
            ge.build()
    import grpc_rdma_amd as g
    from grpc_rdma_amd import stream as gs
    g.init(local_rank_dev)
    from grpc_rdma_amd import shard
    grp = shard.RankGroup(backend="gloo" if EMULATED else "nccl", device=None if EMULATED else device)
    dist = grp.dist

    flags = 2 if args.wire == "direct" else 0
    wl = Workload(g, args.msgs, args.payload)                     # the headline leg's step
    wl_leg = wl if args.leg_msgs == args.msgs else Workload(g, args.leg_msgs, args.payload)   # the comparison legs' step
    workloads = {(args.msgs, args.payload): [wl], (args.leg_msgs, args.payload): [wl_leg]}

    def get_workloads(n_links, msgs_per_link, payload):
        key = (msgs_per_link, payload)
        lst = workloads.setdefault(key, [])
        while len(lst) < n_links:
            lst.append(Workload(g, msgs_per_link, payload))
        return lst[:n_links]

    def barrier():
        dev_sync(torch)
        if dist is not None:
            dist.barrier()
        dev_sync(torch)

    def measure(ring_kb, steps, warmup, verify, instrument, n_links=1, msgs_per_link=None, payload=None,
                pipeline=False, max_sge=None, wire_flags=None, wls=None, reps=1, sends=None, promise=False,
                reindex=False, bidi=False, fused_wire=None):
        """n_links connections with rings of ring_kb KiB: calibrate the number of rounds, capture the graph, time
        `steps` replays.  Then verify, optionally instrument."""
        ring = ring_kb * 1024
        max_sge = max_sge or args.max_sge
        if sends is None:
            sends = args.sends if (pipeline and n_links == 1) else 1
        wf = flags if wire_flags is None else wire_flags
        wls = wls or get_workloads(n_links, msgs_per_link or args.leg_msgs, payload or args.payload)
        links, keep = [], []
        prev = None
        for k, w in enumerate(wls):
            if bidi and k % 2 == 1:
                # BASELINE configs[3] is BIDIRECTIONAL streaming: the second link of a pair runs the other way over the
                # same two ends -- each end sender and receiver at once (pair.cc:264-286), both directions in every launch
                rx, tx = prev
            else:
                tx, rx = g.Pair(ring, max_sge, wf), g.Pair(ring, max_sge, wf)
                g.connect_pairs(tx, rx)
                prev = (tx, rx)
            scap = len(w.lens) * 2 + 64 + w.N // 256
            dst_cap = w.N + 16 * scap + 4096
            dst = g.DeviceBuffer(nbytes=dst_cap)
            links.append((tx, rx, w.sge, dst.ptr, dst_cap, scap))
            keep.append((tx, rx, dst, dst_cap, w))
        w0 = wls[0]
        est_rounds = max(8, 4 * (w0.E // (ring // 2) + 2), 2 * (len(w0.lens) // min(max_sge, 4095) + 2))
        job = gs.MultiStreamJob(links, est_rounds)
        total_n = sum(w.N for w in wls)
        if True:
            job.set_pipeline(pipeline)
            if sends > 1:
                job.set_sends(sends)                   # `sends` consecutive Sends in one plan per round, then one drain
            if promise:
                job.set_promised_credit(True)          # the Send priced with the credit the drain in its launch will post
            if reindex:
                job.set_rebuild_index(True)            # the slice table counts as rewritten: k_tx_index in every step
            if fused_wire is not None:
                job.set_fused_wire(fused_wire)         # few links, small rings: the wire inside the planner pair's launch
            r = job.run(gs.RUN_EAGER)                  # calibration: how many rounds are needed
            assert r.done, "calibration pass did not deliver everything (%d/%d bytes)" % (
                r.bytes_delivered, total_n)
            # (tx_rounds counts Sends)
            rounds = int(max(-(-int(r.tx_rounds) // sends), r.rx_rounds))
            if sends > 2:
                rounds += 2  # (a ring every round fills: later passes start at another ring phase and may need a round more)
            if sends > 2 and pipeline and not promise:
                # a credit-limited ring: the paired graph sees its credit a round late and needs more rounds than the
                # in-order calibration pass -- replay with room, then keep what a pass really used
                job.set_rounds(2 * rounds + 8)
                for _ in range(2):
                    r = job.run(gs.RUN_GRAPH)
                    assert r.done and r.bytes_delivered == total_n
                rounds = int(r.rx_rounds) + 2
            job.set_rounds(rounds)
            r = job.run(gs.RUN_GRAPH)                  # capture + first replay
            assert r.done and r.bytes_delivered == total_n
            wire_groups = job.wire_groups()
            use_streams = args.launch == "streams"
            launch = lambda: job.launch(use_streams)  # noqa: E731
        for _ in range(warmup):
            launch()
        job.sync()
        # `reps` timed regions of exactly `steps` steps each (the reference's protocol: 3 repetitions, the median
        # counts; examples/cpp/micro-bench); every region is bracketed by barrier + synchronize on both sides
        all_elapsed = []
        for _rep in range(max(1, reps)):
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                launch()
            job.sync()
            dev_sync(torch)
            e = time.perf_counter() - t0
            all_elapsed.append(grp.max(e))
            barrier()
        elapsed = sorted(all_elapsed)[len(all_elapsed) // 2]
        out = {"elapsed": elapsed, "all_elapsed": all_elapsed, "rounds": rounds, "sends": sends, "verified": None, "classes": None,
               "wire_groups": wire_groups,
               "user_bytes": sum(w.user_bytes for w in wls), "N": total_n,
               "E": sum(w.E for w in wls)}
        if verify:  # correctness of what the timed region produced (untimed)
            r = job.run(gs.RUN_GRAPH)
            assert r.done and r.bytes_delivered == total_n and r.bytes_sent == total_n
            for li, (tx, rx, dst, dst_cap, w) in enumerate(keep):
                if li not in (0, len(keep) - 1):
                    continue                       # first and last link, byte for byte
                ds = job.delivered_slices(li)
                got = dst.read(dst_cap)
                stream = b"".join(got[o:o + n] for o, n in ds)
                exp = b"".join(w.expected_wire(i) for i in range(w.n_msgs))
                assert stream == exp, "delivered byte stream differs from the framed messages"
                assert rx.ring_mem() == bytes(ring), "ring not zero after the drain"
            out["verified"] = True
        if instrument:  # per-kernel time, HIP events on the launch stream
            inst = None
            for _ in range(3):
                inst = job.run(gs.RUN_INSTRUMENTED)
            classes = {}
            for i, name in enumerate(gs.CLASS_NAMES):
                n = int(inst.launches_class[i])
                if n:
                    classes[name] = {"launches": n, "ms": inst.ms_class[i],
                                     "us_per_launch": 1e3 * inst.ms_class[i] / n}
            out["classes"] = classes
            # the launches of the schedule that was TIMED (pipelined: planner pair; scatter + next gather; wire), each
            # by itself between two events -- same work, same order as the graph, which is a chain
            if pipeline:
                try:
                    inst = None
                    for _ in range(3):
                        inst = job.run(gs.RUN_INSTRUMENTED_SCHEDULE)
                    assert inst.done and inst.bytes_delivered == total_n
                    out["schedule_classes"] = {
                        name: {"launches": int(inst.launches_class[i]), "ms": inst.ms_class[i],
                               "us_per_launch": 1e3 * inst.ms_class[i] / int(inst.launches_class[i])}
                        for i, name in enumerate(gs.CLASS_NAMES) if int(inst.launches_class[i])}
                except Exception as e:  # (a job that is not on the paired schedule: the in-order classes stand alone)
                    out["schedule_classes_error"] = str(e)[:160]
        job.close()
        for tx, rx, dst, _c, _w in keep:
            tx.close(); rx.close(); dst.free()
        return out

    def measure_with_h2(ring_kb, steps, warmup, boundary_step=None, bulk_pairs=None, ticks=False, chunks=None,
                        fused=None):
        """The same step with the HTTP/2 stages INSIDE the timed device pipeline: k_h2_frame_index + k_h2_frame_emit rebuild
        the slice list from the message table, the job carries it through the connection, k_h2_deframe
        parses what was delivered (events: frames, message boundaries, payload pieces).  Two jobs over
        the one connection alternate, so framing / deframing of neighbouring steps run beside a job."""
        from grpc_rdma_amd import h2dev
        ring = ring_kb * 1024
        w = wl_leg
        tx, rx = g.Pair(ring, args.max_sge, flags), g.Pair(ring, args.max_sge, flags)
        g.connect_pairs(tx, rx)
        scap = len(w.lens) * 2 + 64 + w.N // 256
        dst_cap = w.N + 16 * scap + 4096
        msgs = [(w.payload_buf.ptr + i * w.msg_len, w.msg_len, 1, 0) for i in range(w.n_msgs)]
        parser = h2dev.Parser(False, boundary_step=boundary_step, bulk_pairs=bulk_pairs, ticks=ticks, chunks=chunks)
        assert parser.open_streams([1]) == 0      # a client-side parser: the call runs on stream 1
        jobs, pipes, dsts = [], [], []
        env_fused = os.environ.get("GRDMA_H2_PIPE_FUSED")
        if fused is not None:  # (read when a pipe is created) 0: stages enqueued around the job's graph, timed by events
            os.environ["GRDMA_H2_PIPE_FUSED"] = "1" if fused else "0"
        n_pipes = int(os.environ.get("BENCH_H2_PIPES", "2"))
        for _ in range(n_pipes):
            dst = g.DeviceBuffer(nbytes=dst_cap)
            est = max(8, 4 * (w.E // (ring // 2) + 2), 2 * (len(w.lens) // min(args.max_sge, 4095) + 2))
            job = gs.MultiStreamJob([(tx, rx, w.sge, dst.ptr, dst_cap, scap)], est)
            job.set_pipeline(bool(args.pipeline))
            h2_sends = args.sends if args.pipeline else 1
            if h2_sends > 1:
                job.set_sends(h2_sends)
            r = job.run(gs.RUN_EAGER)
            job.set_rounds(int(max(-(-int(r.tx_rounds) // h2_sends), r.rx_rounds)))
            r = job.run(gs.RUN_GRAPH)
            assert r.done and r.bytes_delivered == w.N
            delivered = len(job.delivered_slices(0))
            pipes.append(h2dev.Pipe(job, msgs, parser, delivered, 4 * len(w.lens) + 1024))
            jobs.append(job)
            dsts.append(dst)
        if fused is not None:
            if env_fused is None:
                os.environ.pop("GRDMA_H2_PIPE_FUSED", None)
            else:
                os.environ["GRDMA_H2_PIPE_FUSED"] = env_fused
        for i in range(max(2, warmup)):
            pipes[i % n_pipes].enqueue(False)
        for p_ in pipes:
            p_.sync()
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            pipes[i % n_pipes].enqueue(False)
        for p_ in pipes:
            p_.sync()
        torch.cuda.synchronize()
        elapsed = grp.max(time.perf_counter() - t0)
        barrier()
        ok = True
        stage_us = {}
        for p_ in pipes:  # what the last step of each pipe produced
            r = p_.sync(want_events=True)
            stage_us = {"frame_us": r["frame_us"], "deframe_us": r["deframe_us"], "delivered_slices": p_.delivered,
                        "events": r["events"], "bulk_steps": r["bulk_steps"], "bulk_frames": r["bulk_frames"],
                        "boundary_steps": r["boundary_steps"],
                        "deframe_ticks": {"wait_for_windows": r["t_wait"], "bulk_steps": r["t_bulk"],
                                          "boundary_steps": r["t_boundary"],
                                          "bytewise_path": r["t_serial"], "total": r["t_total"]}}
            evs = r["event_list"]
            ok = ok and r["h2_error"] == 0 and not r["frame_overflow"] and not r["deframe_overflow"] and \
                r["framed"] == len(w.lens) and r["parsed"] == p_.delivered and \
                sum(1 for e in evs if e[0] == 5) == w.n_msgs and sum(e[2] for e in evs if e[0] == 4) == w.n_msgs * w.msg_len
        # the events of the last step once more, through a parser of its own with the one-wave sequential deframer over the
        # same delivered slices (both are product paths: the chunked deframer must not change a single event)
        try:
            if chunks is not False:
                ps_ = h2dev.Parser(False, boundary_step=boundary_step, bulk_pairs=bulk_pairs, chunks=False)
                assert ps_.open_streams([1]) == 0
                last = (steps - 1) % n_pipes if steps else n_pipes - 1
                rl = pipes[last].sync(want_events=True)
                err_, ev_seq = ps_.deframe(dsts[last].ptr, jobs[last].delivered_slices(0), cap=pipes[last].events_cap)
                ps_.close()
                stage_us["events_equal_sequential_deframer"] = bool(err_ == 0 and ev_seq == rl["event_list"])
                ok = ok and stage_us["events_equal_sequential_deframer"]
        except Exception as e:
            stage_us["events_equal_sequential_deframer_error"] = err_text(e)
        if os.environ.get("BENCH_H2_CHUNK_PHASES"):
            rows = parser.chunk_phases()
            live = [r for r in rows[:-1] if r[0]]
            ph = [[(r[i] - r[0]) for i in range(5)] + [r[5]] for r in live]  # (every XCD has a clock of its own)
            mg = rows[-1]
            import statistics as st_
            sys.stderr.write("chunk phases, device-clock ticks from each chunk's own start (the deframer's 779 us are 1.87 M ticks):\n")
            for name, i in (("cuts", 1), ("copied", 2), ("parsed", 3), ("compared", 4), ("slices", 5)):
                col = [p_[i] for p_ in ph]
                sys.stderr.write("  %-9s min %7d median %7d max %7d\n" % (name, min(col), int(st_.median(col)), max(col)))
            sys.stderr.write("  merge (workgroup 0): verified %d copied %d end %d\n" % tuple(m_ - mg[0] for m_ in mg[1:4]))
        planned, merged = parser.chunk_stats()
        stage_us["deframe_calls"] = max(2, warmup) + steps
        stage_us["deframe_calls_planned_over_chunks"] = planned
        stage_us["deframe_calls_merged_from_chunks"] = merged
        for p_ in pipes:
            p_.close()
        for j_ in jobs:
            j_.close()
        parser.close()
        tx.close(); rx.close()
        for d_ in dsts:
            d_.free()
        return {"elapsed": elapsed, "verified": ok, "stages": stage_us}

    if args.h2_only:  # (internal) the with-h2 legs alone: chunked deframer (default) and the sequential one
        o_ = {}
        for tag, ch, fu in (("value_with_h2", None, None), ("value_with_h2_stages_around_the_graph", None, False),
                            ("value_with_h2_sequential_deframer", False, False)):
            if tag != "value_with_h2" and os.environ.get("BENCH_H2_DEFAULT_LEG_ONLY"):
                continue
            h2_ = measure_with_h2(args.ring_kb, args.steps, max(2, args.warmup), chunks=ch,
                                  fused=fu)
            o_[tag] = round(wl.user_bytes * args.steps * world / h2_["elapsed"] / (1 << 30), 3)
            o_[tag + "_verified"] = h2_["verified"]
            o_[tag + "_stages"] = h2_["stages"]
        print(json.dumps(o_))
        return

    if args.h2_bulk_pairs_only:  # the child of the value_with_h2_bulk_pairs leg: one leg, one JSON line
        few = max(2, args.steps // 4)
        h2_ = measure_with_h2(args.ring_kb, few, 2, bulk_pairs=False)
        print(json.dumps({"value_with_h2_bulk32": round(wl.user_bytes * few * world / h2_["elapsed"] / (1 << 30), 3),
                          "with_h2_bulk32_deframe_us": h2_["stages"]["deframe_us"],
                          "with_h2_bulk32_verified": h2_["verified"]}))
        return

    schedule = "pipelined" if args.pipeline else "sequential"
    head = None
    if args.pipeline:
        try:
            head = measure(args.ring_kb, args.steps, args.warmup, not args.no_verify, True, pipeline=True, reps=args.reps,
                           promise=args.promise, wls=[wl])
        except Exception as e:  # keep the line: fall back to the plain schedule and say so
            schedule = "sequential (pipelined run failed: %s)" % str(e)[:120]
    seq = None
    if head is None or not args.no_extra_legs:
        seq = measure(args.ring_kb, args.steps if head is None else max(2, args.steps // 2),
                      args.warmup if head is None else 1, not args.no_verify, head is None, pipeline=False,
                      reps=args.reps if head is None else 1, wls=[wl] if head is None else None)
    if head is None:
        head, seq = seq, None
    graph_head = head
    elapsed, rounds, classes, verified = head["elapsed"], head["rounds"], head["classes"], head["verified"]
    ring = args.ring_kb * 1024
    small = None
    if args.ring_kb != 4096 and not args.no_small_ring:
        # (round 6: the paired schedule with the promised credit, and -- one link, a small ring -- the wire in the planner
        #  pair's launch; the sequential schedule, what this leg ran through round 5, beside it)
        small = measure(4096, max(2, args.steps // 2), 1, not args.no_verify, False, pipeline=True, sends=1, promise=True)
        small_seq = measure(4096, max(2, args.steps // 2), 1, False, False)

    # algorithmic bytes per step per kernel class:
    alg = {"gather": 2 * wl.N,          # K1: N read + N written (tags E-N by tx_plan)
           "wire": 2 * wl.E,            # loop-back stand-in for the NIC: E read + E written
           "rx_apply": 3 * wl.N}        # K4: N read + N written + N cleared (tags by rx_plan)
    dom = max((k for k in classes if k in alg), key=lambda k: classes[k]["ms"])
    per_launch = alg[dom] / classes[dom]["launches"]
    achieved = per_launch / (classes[dom]["us_per_launch"] * 1e-6) / 1e9
    kname = "k_rx_apply" if dom == "rx_apply" else "k_copy"
    traffic = None
    pmc, pmc_d, pmc_stale = pmc_summary_for(args.ring_kb)
    # (the summary's figure is per FULL-SIZE launch -- two Sends of 4095 slices scattered, two gathered: the same launch
    #  whatever the number of messages per step)
    pmc_usable = pmc_d is not None and args.wire == "staged" and args.max_sge == 4095 and args.sends == 2 and args.payload == MIB
    if pmc_usable:
        try:
            k = pmc_d["kernels"][kname]
            # one FULL launch against the algorithmic bytes of one full launch (the plain mean also counts the
            # short last round of a step and the warm-ups)
            traffic = k.get("hbm_traffic_bytes_full_size_launch", k["hbm_traffic_bytes"])
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": "%s (%s)" % (kname, dom),
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "traffic_source": (os.path.relpath(pmc, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                   "command, committed, collected from these sources: sources_sha16 %s; NOT re-measured "
                                   "in this run)" % sources_sha16()) if traffic is not None
                else ("none: " + (pmc_stale or "not the profiled configuration")),
                "bytes_per_launch": int(per_launch),
                "us_per_launch": round(classes[dom]["us_per_launch"], 2)}
    # The default schedule runs the scatter of round t and the gather of round t + 1 as ONE launch (k_rx_apply_gather):
    # that launch is the dominant HBM kernel of the timed region.  Its algorithmic bytes: 3 x the payload of the
    # rounds it scatters (read + written + cleared) + 2 x the payload of the rounds it gathers (read + written); the
    # rounds are the slice list in chunks of max_sge slices (no Send is cut by credit at this ring size).
    sched = head.get("schedule_classes") or {}
    if "scatter_gather" in sched and args.schedule != "engine":
        sge = min(args.max_sge, 4095) * head.get("sends", 1)  # (slices per round)
        chunks = [sum(wl.lens[i:i + sge]) for i in range(0, len(wl.lens), sge)]
        sg = sched["scatter_gather"]
        if len(chunks) == rounds and sg["launches"] == rounds - 1:
            fused_bytes = 3 * sum(chunks[:-1]) + 2 * sum(chunks[1:])
        else:
            fused_bytes = 5 * wl.N * sg["launches"] / max(1, rounds)
        per_launch_sg = fused_bytes / sg["launches"]
        ach = per_launch_sg / (sg["us_per_launch"] * 1e-6) / 1e9
        traffic_sg = None
        if pmc_usable:
            try:
                k = pmc_d["kernels"]["k_rx_apply_gather"]
                traffic_sg = k.get("hbm_traffic_bytes_full_size_launch")
            except Exception:
                traffic_sg = None
        alone = {"kernel": roofline["kernel"], "achieved": roofline["achieved"], "frac": roofline["frac"],
                 "bytes_per_launch": roofline["bytes_per_launch"], "us_per_launch": roofline["us_per_launch"],
                 "traffic": roofline["traffic"],
                 "note": "the scatter as a launch of its own (in-order instrumented pass; what rounds 1 and 2 reported)"}
        roofline = {"bound": "hbm",
                    "kernel": "k_rx_apply_gather (scatter of round t + gather of round t + 1: one launch of the timed schedule)",
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": traffic_sg,
                    "traffic_source": (os.path.relpath(pmc, ROOT) + " (hbm_traffic_bytes_full_size_launch: rocprofv3 --pmc "
                                       "FETCH_SIZE / WRITE_SIZE passes of this command, mean over the full-size launches "
                                       "-- %d B algorithmic for one of those; committed, collected from these sources: "
                                       "sources_sha16 %s; NOT re-measured in this run)"
                                       % (5 * max(chunks), sources_sha16())) if traffic_sg is not None
                    else ("none: " + (pmc_stale or "not the profiled configuration")),
                    "bytes_per_launch": int(per_launch_sg), "us_per_launch": round(sg["us_per_launch"], 2),
                    "launches_per_step": sg["launches"],
                    "measured": "HIP events around every launch of the timed schedule, enqueued in the graph's order on "
                                "one stream (GRDMA_RUN_INSTRUMENTED_SCHEDULE); mean over the %d fused launches of a step, "
                                "the last of which gathers the short last round" % sg["launches"],
                    "scatter_alone": alone}
        tot_s = sum(v["ms"] for v in sched.values())
        roofline["schedule_kernels"] = {k: {"launches": v["launches"], "us_per_launch": round(v["us_per_launch"], 2),
                                            "share_of_kernel_time": round(v["ms"] / tot_s, 3)} for k, v in sched.items()}
    # what the roofline of the dominant HBM kernel does not show: the kernel that takes the most TIME (a planner is
    # latency-bound and moves next to nothing), and the step as a whole against the HBM peak
    tot_ms = sum(v["ms"] for v in classes.values())
    top = max(classes, key=lambda k: classes[k]["ms"])
    step_bytes = 2 * wl.N + 3 * wl.E  # K1 (N read + E written) + K4 (E read + N written + E cleared); the wire is the NIC's
    step_s = elapsed / args.steps
    roofline["dominant_by_time"] = {
        "kernel": top, "us_per_launch": round(classes[top]["us_per_launch"], 2),
        "share_of_kernel_time": round(classes[top]["ms"] / tot_ms, 3),
        "planner_share_of_kernel_time": round(sum(classes[k]["ms"] for k in ("tx_plan", "rx_plan") if k in classes) / tot_ms, 3),
        "note": "per-class HIP-event time of the instrumented in-order pass (a class = the launches between two events)"}
    if roofline.get("schedule_kernels"):
        # the same question for the schedule that was timed: its launches are the planner pair (drain plan of round t +
        # send plan of round t + 1), the fused scatter + gather and the wire
        sk = roofline["schedule_kernels"]
        top_s = max(sk, key=lambda k: sk[k]["share_of_kernel_time"])
        roofline["dominant_by_time_in_order_pass"] = roofline["dominant_by_time"]
        roofline["dominant_by_time"] = {
            "kernel": top_s + {"plan_pair": " (k_plan_pair_mw)", "scatter_gather": " (k_rx_apply_gather)"}.get(top_s, ""),
            "us_per_launch": sk[top_s]["us_per_launch"], "share_of_kernel_time": sk[top_s]["share_of_kernel_time"],
            "planner_share_of_kernel_time": round(sum(sk[k]["share_of_kernel_time"] for k in ("tx_plan", "plan_pair", "rx_plan")
                                                      if k in sk), 3),
            "note": "per-launch HIP-event time of the timed schedule's launches (GRDMA_RUN_INSTRUMENTED_SCHEDULE)"}
    roofline["step_level"] = {"bytes": int(step_bytes), "achieved": round(step_bytes / step_s / 1e9, 1), "unit": "GB/s",
                              "frac": round(step_bytes / step_s / 1e9 / HBM_PEAK_GBPS, 4),
                              "frac_with_wire": round((step_bytes + 2 * wl.E) / step_s / 1e9 / HBM_PEAK_GBPS, 4)}
    try:
        roofline["measured_ceiling"] = None if EMULATED else measured_copy_ceiling(torch, device)
    except Exception as e:
        roofline["measured_ceiling"] = {"error": str(e)[:120]}

    total_user = wl.user_bytes * args.steps * world
    value = total_user / elapsed / (1 << 30)
    out = {
        "metric": "streaming GiB/s @1 MiB msgs (client-streaming, 1 connection per GPU)",
        "value": round(value, 3), "unit": "GiB/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "repetitions": {"n": len(head.get("all_elapsed", [elapsed])), "reported": "median",
                        "ms_per_step": [round(1e3 * e / args.steps, 4) for e in head.get("all_elapsed", [elapsed])]},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic" if not EMULATED else "synthetic; EMULATED DEVICE (GRDMA_BENCH_EMULATED=1: the product sources on the "
                                                 "wave emulator, CPU tensors, gloo) -- a dry run of the script, `value` is NOT a measurement",
        "value_is": "device-resident: slices in HBM before the timed region, delivered slices left in HBM "
                    "(host-slice rate through the endpoint vtable: value_endpoint_vtable)",
        "config": {"workload": "client-streaming 1 MiB payloads, 1 connection on 1xMI355X "
                               "(BASELINE.json configs[2])",
                   "msgs_per_step": args.msgs, "slices_per_msg": wl.slices_per_msg,
                   "ring_kib": args.ring_kb, "max_sge": args.max_sge, "wire": args.wire,
                   "knobs_note": ("the ring size is the reference's own knob (GRPC_RDMA_RING_BUFFER_SIZE_KB; its bandwidth "
                                  "plots sweep it); max_sge %d is how many slices (ring records) ONE Send takes.  In the "
                                  "reference that count is the HCA's scatter-gather limit (30 on mlx5: its work request "
                                  "gathers the records, pair.cc:53-60, 676-734); here the device has gathered them into the "
                                  "contiguous staging buffer and the NIC wire posts num_sge = 1 per work request "
                                  "(csrc/grdma_wire_verbs.cc: <= 2 requests per Send whatever max_sge is), so max_sge only "
                                  "places the cut between Sends and no HCA limit stands against this value -- it is a "
                                  "legal setting of GRPC_RDMA_MAX_SGE, not the reference's default.  The same workload at "
                                  "the reference's DEFAULT knobs (4 MiB ring, "
                                  "max_sge 30) is value_ring4096_sge30 -- an order of magnitude below `value` -- with the "
                                  "reference's CPU codec at those knobs beside it (cpu_baseline_ring4096_sge30)"
                                  % args.max_sge),
                   "schedule": schedule,
                   # (the spread of the timed regions, where the driver's parser keeps it: `value` is their median)
                   "repetitions_ms_per_step": [round(1e3 * e / args.steps, 4) for e in head.get("all_elapsed", [elapsed])],
                   "sends_per_round": head.get("sends", 1),
                   "sends_per_round_note": "a round = rdma_flush's loop while the ring has room (rdma_bp_posix.cc:470-524): "
                                           "that many Sends of <= max_sge slices back to back, then the peer's endpoint reads "
                                           "until one would block.  Two Sends are 63 MiB: the ring (256 MiB) holds four such rounds, "
                                           "as the 128 MiB ring of the earlier rounds held four rounds of one Send -- that "
                                           "configuration is value_ring128m_one_send_per_round",
                   "slice_index": "`value` replays ONE slice table: the index its Sends are priced from (k_tx_index, prefix sums over "
                                  "the table) is built in the job's first step only.  A real stream brings a new slice buffer with "
                                  "every grpc_endpoint_write (rdma_bp_posix.cc:559-586): value_index_rebuilt_every_step is the same "
                                  "step with the index rebuilt in EVERY step",
                   "rounds_per_step": rounds, "connections_per_gpu": 1,
                   "stages": "gather+encode, wire, ready-detect, decode+scatter+zero, credit"},
        "roofline": roofline,
        "kernels": {k: {"launches": v["launches"], "us_per_launch": round(v["us_per_launch"], 2)}
                    for k, v in classes.items()},
        "endpoint_bytes_per_step": wl.N, "ring_bytes_per_step": wl.E, "verified": verified,
    }
    # ---- single-stream fan-out (BASELINE.json configs[4]): the one collective of the path ---
    # (on ONE GPU the scatter step is empty -- the ingest rank keeps its share in place -- but the leg still ingests the
    #  128 MiB stream and checksums the delivered arena against the framed messages: the checker runs on hardware)
    if not args.no_fanout and (world > 1 or not args.no_extra_legs):
        try:
            fo = fanout_leg(g, gs, grp, torch, args, flags, n_msgs=args.fanout_msgs)
            if rank == 0:
                out.update(fo)
        except Exception as e:
            out["fanout_error"] = str(e)
    # ---- unary 64 B ping-pong (BASELINE.json configs[1]): second half of the metric -------
    if not args.no_rtt:
        if rank == 0:  # (one connection on one GPU: rank 0 measures it, in helper processes)
            out.update(rtt_subprocess("--rtt-only", args.rtt_iters, 240, key="rtt_error"))
            if world == 1:
                out.update(rtt_subprocess("--rtt-commands-only", max(1000, args.rtt_iters // 10), key="rtt_read_commands_error"))
    if not args.no_extra_legs and rank == 0:
        try:
            out["conn_setup_us"] = conn_setup_us(g)
        except Exception as e:
            out["conn_setup_error"] = str(e)[:200]
    if not args.no_extra_legs and args.schedule != "engine":
        # the headline step with the slice table counted as rewritten between steps: k_tx_index inside every timed step
        try:
            ri = measure(args.ring_kb, max(2, args.steps // 2), 1, not args.no_verify, False, pipeline=bool(args.pipeline), reindex=True, wls=[wl])
            out["value_index_rebuilt_every_step"] = round(wl.user_bytes * max(2, args.steps // 2) * world / ri["elapsed"] / (1 << 30), 3)
            out["index_rebuilt_every_step_verified"] = ri["verified"]
        except Exception as e:
            out["index_rebuilt_every_step_error"] = str(e)[:200]
    if not args.no_extra_legs and args.leg_msgs != args.msgs:
        # the step of rounds 1 - 5: 256 messages = four rounds of 63 and a fifth of four, with its own first gather and last
        # scatter -- what `value` was quoted on until round 5 (same kernels, same schedule, same ring)
        try:
            o256 = measure(args.ring_kb, args.steps, 2, not args.no_verify, False, pipeline=bool(args.pipeline), wls=[wl_leg])
            out["value_msgs%d_per_step" % args.leg_msgs] = round(wl_leg.user_bytes * args.steps * world / o256["elapsed"] / (1 << 30), 3)
            out["rounds_per_step_msgs%d" % args.leg_msgs] = o256["rounds"]
            out["msgs%d_per_step_verified" % args.leg_msgs] = o256["verified"]
        except Exception as e:
            out["msgs%d_per_step_error" % args.leg_msgs] = str(e)[:200]
    if not args.no_extra_legs and args.leg_msgs == 256 and args.sends == 2 and args.max_sge == 4095:
        # A step of 256 messages is four rounds of 63 (two Sends of <= 4095 slices: 31 + 32 messages of 130) and a FIFTH of
        # four messages -- wire, planner pair and scatter launched once more for 1.5 % of the bytes.  A continuous stream has
        # no such round; the same step with 252 messages (four full rounds) shows what it costs `value`.
        try:
            fw = Workload(g, 252)
            fr = measure(args.ring_kb, max(2, args.steps // 2), 1, not args.no_verify, False, pipeline=bool(args.pipeline), wls=[fw])
            out["value_msgs252_full_rounds_only"] = round(fw.user_bytes * max(2, args.steps // 2) * world / fr["elapsed"] / (1 << 30), 3)
            out["rounds_per_step_msgs252"] = fr["rounds"]
            out["msgs252_verified"] = fr["verified"]
        except Exception as e:
            out["msgs252_error"] = str(e)[:200]
    if not args.no_extra_legs:
        # the same headline step with PRNG payload bytes (seed 1234): the reference's second payload kind
        try:
            pw = Workload(g, args.msgs, prng_seed=1234)
            pr = measure(args.ring_kb, max(2, args.steps // 2), 1, not args.no_verify, False, pipeline=bool(args.pipeline), wls=[pw])
            out["value_prng_payload"] = round(pw.user_bytes * max(2, args.steps // 2) * world / pr["elapsed"] / (1 << 30), 3)
            out["prng_payload_verified"] = pr["verified"]
        except Exception as e:
            out["prng_payload_error"] = str(e)[:200]
    if not args.no_extra_legs and args.pipeline and args.schedule != "engine" and (args.sends > 1 or args.ring_kb != 131072):
        # the configuration rounds 1 - 3 and the first half of round 4 reported as `value`: a 128 MiB ring, ONE Send per
        # round (with two, a round is 63 MiB and that ring holds two of them: every other round waits for its credit)
        try:
            o1 = measure(131072, max(2, args.steps // 2), 1, not args.no_verify, False, pipeline=True, sends=1)
            out["value_ring128m_one_send_per_round"] = round(wl_leg.user_bytes * max(2, args.steps // 2) * world / o1["elapsed"] / (1 << 30), 3)
            out["rounds_per_step_ring128m_one_send_per_round"] = o1["rounds"]
        except Exception as e:
            out["ring128m_one_send_per_round_error"] = str(e)[:200]
        # ... and two Sends per round AT that 128 MiB ring, which holds two such rounds: with the promised credit
        # (grdma_stream_job_set_promised_credit) the Send of round t + 1 is priced with the credit the drain of round t is
        # about to post, so two rounds of ring are enough (the planner pair's launch then runs drain plan and send plan
        # one after the other)
        try:
            o2 = measure(131072, max(2, args.steps // 2), 1, not args.no_verify, False, pipeline=True, sends=2, promise=True)
            out["value_ring128m_promised_credit"] = round(wl_leg.user_bytes * max(2, args.steps // 2) * world / o2["elapsed"] / (1 << 30), 3)
            out["rounds_per_step_ring128m_promised_credit"] = o2["rounds"]
        except Exception as e:
            out["ring128m_promised_credit_error"] = str(e)[:200]
    if seq is not None:  # same workload and ring, five kernels per round strictly in order
        out["value_sequential"] = round(
            wl_leg.user_bytes * max(2, args.steps // 2) * world / seq["elapsed"] / (1 << 30), 3)
    if not args.no_extra_legs and args.wire == "staged":
        # GRDMA_WIRE_DIRECT: the gather writes the records straight into the peer ring
        # (HBM / xGMI peer memory), no staging copy and no wire kernel
        try:
            dr = measure(args.ring_kb, args.steps, max(2, args.warmup), not args.no_verify, False,
                         pipeline=bool(args.pipeline), wire_flags=2)
            out["value_wire_direct"] = round(wl_leg.user_bytes * args.steps * world / dr["elapsed"] / (1 << 30), 3)
            out["wire_direct_verified"] = dr.get("verified")
            out["config"]["wire_direct_leg"] = ("the same step with GRDMA_WIRE_DIRECT pairs: the gather writes the records straight into the peer "
                                                "ring (HBM / xGMI peer memory), no staging copy, no wire kernel -- two launches per round "
                                                "(planner pair; scatter + next gather), same schedule and verification as the headline")
        except Exception as e:
            out["wire_direct_error"] = str(e)[:200]
    half = max(2, args.steps // 2)
    if not args.no_extra_legs or os.environ.get("BENCH_H2"):
        # frame -> endpoint -> deframe, all three inside the timed device pipeline
        try:
            hh = measure_with_h2(args.ring_kb, args.steps, max(2, args.warmup))
            out["value_with_h2"] = round(wl_leg.user_bytes * args.steps * world / hh["elapsed"] / (1 << 30), 3)
            out["with_h2_verified"] = hh["verified"]
            out["with_h2_stages"] = hh["stages"]
            out["config"]["with_h2_leg"] = ("k_h2_frame_index + k_h2_frame_emit -> the job -> k_h2_deframe inside the timed pipeline, library defaults "
                                            "(framing and deframing are kernel nodes of the job's own graph: one launch per step; "
                                            "message-boundary step on, 64 frames per bulk step, the delivered slices parsed as up to 256 "
                                            "chunks side by side and merged after the chain of end states verified, no clock samples); "
                                            "value_with_h2_stages_around_the_graph: the stages enqueued around the job's graph launch; "
                                            "value_with_h2_sequential_deframer: one parsing wave over the whole list; "
                                            "value_with_h2_no_boundary_step: message starts byte-wise; "
                                            "value_with_h2_bulk32: 32 frames per bulk step (GRDMA_H2_BULK_PAIRS=0)")
        except Exception as e:
            out["with_h2_error"] = err_text(e)
        few = max(2, args.steps // 4)
        try:  # per-stage times: the stages enqueued around the job's graph, each between two events
            hs = measure_with_h2(args.ring_kb, few, 2, fused=False)
            out["value_with_h2_stages_around_the_graph"] = round(wl_leg.user_bytes * few * world / hs["elapsed"] / (1 << 30), 3)
            for k_ in ("frame_us", "deframe_us"):
                out["with_h2_stages"][k_] = hs["stages"][k_]
            out["with_h2_stages"]["stage_times_from"] = ("a run with GRDMA_H2_PIPE_FUSED=0 (stages enqueued around the job's "
                                                         "graph, event pairs around each; includes the graph boundary)")
        except Exception as e:
            out["with_h2_unfused_error"] = err_text(e)
        try:  # the phase ticks of the deframing kernel (a short run with the clock samples on)
            ht = measure_with_h2(args.ring_kb, 2, 2, ticks=True, fused=False)
            out["with_h2_stages"]["deframe_ticks"] = ht["stages"]["deframe_ticks"]
            out["with_h2_stages"]["deframe_us_with_clock_samples"] = ht["stages"]["deframe_us"]
        except Exception as e:
            out["with_h2_ticks_error"] = err_text(e)
        try:  # the same leg with message starts left to the byte-wise automaton
            h0 = measure_with_h2(args.ring_kb, few, 2, boundary_step=False)
            out["value_with_h2_no_boundary_step"] = round(wl_leg.user_bytes * few * world / h0["elapsed"] / (1 << 30), 3)
            out["with_h2_no_boundary_step_deframe_us"] = h0["stages"]["deframe_us"]
        except Exception as e:
            out["with_h2_no_boundary_step_error"] = err_text(e)
        try:  # the same leg with the one-wave sequential deframer (no chunks: the default until the end of round 3)
            h1 = measure_with_h2(args.ring_kb, few, 2, chunks=False, fused=False)
            out["value_with_h2_sequential_deframer"] = round(wl_leg.user_bytes * few * world / h1["elapsed"] / (1 << 30), 3)
            out["with_h2_sequential_deframer_deframe_us"] = h1["stages"]["deframe_us"]
        except Exception as e:
            out["with_h2_sequential_deframer_error"] = err_text(e)
        try:  # ... and with 32 frames per bulk step (the default until round 3), in a helper process, N=1 only
            if rank == 0 and world == 1:
                cmd = [sys.executable, os.path.abspath(__file__), "--h2-bulk-pairs-only", "--steps", str(args.steps),
                       "--ring-kb", str(args.ring_kb), "--msgs", str(args.leg_msgs), "--leg-msgs", str(args.leg_msgs),
                       "--schedule", args.schedule]
                r_ = run_json(cmd, 150)
                if "value_with_h2_bulk32" in r_:
                    out.update(r_)
                else:
                    out["with_h2_bulk32_error"] = str(r_.get("error"))[:300]
        except Exception as e:
            out["with_h2_bulk32_error"] = err_text(e)
    if not args.no_extra_legs:
        # the reference's default knobs (4 MiB ring, max_sge 30: rdma_utils.h / config.cc), same workload,
        # with the CPU codec timed at the SAME knobs beside it
        # Rounds = rdma_flush's loop: Sends of <= 30 slices until the ring is full (grdma_stream_job_set_sends(64)), priced
        # as ONE cut of the slice table's index by the small planner workgroups, the drain predicted from the sizes they
        # leave; SEQUENTIAL schedule (five launches per round): every round fills the ring, and the paired schedule would
        # see its credit a round late.
        for key, wf, pl, fw in (("value_ring4096_sge30", None, True, None), ("value_ring4096_sge30_wire_in_its_own_launch", None, True, False),
                                ("value_ring4096_sge30_sequential", None, False, None), ("value_ring4096_sge30_wire_direct", 2, False, None)):
            try:
                # staged wire: the PAIRED schedule with the promised credit -- the Send of round t + 1 waits, inside the
                # planner pair's launch, for the drain plan of round t and is priced with the credit it will post: every
                # round fills the ring, as on the sequential schedule -- and (round 6) with the WIRE of round t in that same
                # launch: wire workgroups in front of the drain's, which wait for them before they look at the ring; two
                # launches per round (planner pair + wire; scatter + next gather).  "_wire_in_its_own_launch": the same
                # with the wire as a k_copy launch (three launches per round, what every job with more links or bigger
                # rings runs).  Direct wire: sequential (the gather of round t + 1 writes the ring, it cannot share a
                # launch with the scatter of round t).
                rk = measure(4096, half, 1, not args.no_verify, False, max_sge=30, pipeline=pl, sends=64, wire_flags=wf, promise=pl,
                             fused_wire=fw)
                out[key] = round(wl_leg.user_bytes * half * world / rk["elapsed"] / (1 << 30), 3)
                out["rounds_per_step_" + key[6:]] = rk["rounds"]
                if key == "value_ring4096_sge30":
                    out["config"]["ring4096_sge30_leg"] = (
                        "4 MiB ring, max_sge 30 (the reference's defaults); a round = Sends of 30 slices until the ring is full "
                        "(grdma_stream_job_set_sends), priced as one cut of the slice table's index; paired schedule with the "
                        "promised credit (grdma_stream_job_set_promised_credit), the wire of a round inside the planner pair's "
                        "launch (%d wire workgroups, grdma_stream_job_wire_groups): two launches per round, %d rounds" % (
                            rk["wire_groups"], rk["rounds"]))
            except Exception as e:
                out[key[6:] + "_error"] = err_text(e)
        try:  # (one Send of 30 slices per round, drained at once: what the reference's loop is without rdma_flush's retries)
            rk1 = measure(4096, 2, 1, False, False, max_sge=30)
            out["value_ring4096_sge30_one_send_per_round"] = round(wl_leg.user_bytes * 2 * world / rk1["elapsed"] / (1 << 30), 3)
        except Exception as e:
            out["ring4096_sge30_one_send_per_round_error"] = err_text(e)
        # mixed message sizes (examples/cpp/test/common.h: uniform in [1, 4 MiB - 1 KiB]), 64 messages per step
        try:
            mw = MixedWorkload(g, 64)
            # (the headline's two Sends per round: the size table a round leaves for its drain holds two Sends' worth, 8192
            #  records, since round 6 -- csrc/grdma_rx_hint.h; one Send per round, what rounds 4-5 ran, beside it)
            mx = measure(args.ring_kb, half, 1, not args.no_verify, False, pipeline=bool(args.pipeline), wls=[mw], sends=args.sends)
            mx1 = measure(args.ring_kb, half, 1, False, False, pipeline=bool(args.pipeline), wls=[mw], sends=1)
            out["value_mixed_sizes_one_send_per_round"] = round(mw.user_bytes * half * world / mx1["elapsed"] / (1 << 30), 3)
            out["value_mixed_sizes"] = round(mw.user_bytes * half * world / mx["elapsed"] / (1 << 30), 3)
            # at the reference's default knobs: the paired schedule with the promised credit and the wire in the planner
            # pair's launch, as the 1 MiB leg runs (round 6; the sequential schedule, what rounds 4-5 reported, beside it)
            mx2 = measure(4096, half, 1, not args.no_verify, False, max_sge=30, wls=[mw], pipeline=True, sends=64, promise=True)
            out["value_mixed_sizes_ring4096_sge30"] = round(mw.user_bytes * half * world / mx2["elapsed"] / (1 << 30), 3)
            out["rounds_per_step_mixed_sizes_ring4096_sge30"] = mx2["rounds"]
            mx3 = measure(4096, half, 1, not args.no_verify, False, max_sge=30, wls=[mw], pipeline=False, sends=64)
            out["value_mixed_sizes_ring4096_sge30_sequential"] = round(mw.user_bytes * half * world / mx3["elapsed"] / (1 << 30), 3)
            out["config"]["mixed_sizes_leg"] = "64 messages, sizes uniform in [1, 4 MiB - 1 KiB] (seed 0), %d MiB per step, %d slices" % (
                mw.user_bytes >> 20, len(mw.lens))
        except Exception as e:
            out["mixed_sizes_error"] = err_text(e)
    if rank == 0 and world == 1 and not args.no_extra_legs:
        # host slices through the endpoint vtable (grpc_endpoint_write / _read, include/grdma_endpoint.hpp):
        # what a gRPC maintainer's process sees, PCIe both ways included
        env = dict(os.environ, GRPC_RDMA_RING_BUFFER_SIZE_KB=str(args.ring_kb), GRPC_PLATFORM_TYPE="RDMA_BP")
        ev = run_json([os.path.join(ROOT, "tools", "endpoint_stream"), "1024", str(MIB), "1", "0", "2"], 120, env)
        out["value_endpoint_vtable"] = ev.get("GiBps")
        out["endpoint_vtable"] = ev
        # what the link gives a copy engine in both directions at once (tools/pcie_probe.py): the ceiling of a stream that
        # crosses it once each way
        pc = run_json([sys.executable, os.path.join(ROOT, "tools", "pcie_probe.py"), "256"], 120, env)
        out["pcie_ceiling"] = pc
        if pc.get("both_each_GiBps") and ev.get("GiBps"):
            out["value_endpoint_vtable_frac_of_pcie_ceiling"] = round(ev["GiBps"] / pc["both_each_GiBps"], 3)
        # ... every write a chain of its own (round 4's behaviour: no coalescing in the send buffer that waits)
        evc = run_json([os.path.join(ROOT, "tools", "endpoint_stream"), "1024", str(MIB), "1", "0", "2"], 120,
                       dict(env, GRPC_RDMA_HIP_COALESCE="0"))
        out["value_endpoint_vtable_no_coalescing"] = evc.get("GiBps")
        # the same at the reference's default ring (GRPC_RDMA_RING_BUFFER_SIZE_KB 4096, config.cc): ring and receive
        # windows stay cache- and IOMMU-resident
        ev4 = run_json([os.path.join(ROOT, "tools", "endpoint_stream"), "1024", str(MIB), "1", "0", "2"], 120,
                       dict(env, GRPC_RDMA_RING_BUFFER_SIZE_KB="4096"))
        out["value_endpoint_vtable_ring4096"] = ev4.get("GiBps")
        # ... and with the endpoint's send buffers off (every write outstanding until its last Send, the reference's flow)
        ev0 = run_json([os.path.join(ROOT, "tools", "endpoint_stream"), "1024", str(MIB), "1", "0", "2"], 120,
                       dict(env, GRPC_RDMA_HIP_SEND_BUFFER_KB="0"))
        out["value_endpoint_vtable_no_send_buffers"] = ev0.get("GiBps")
        # ... and without the reader's byte-sum check of every delivered slice (~57 us of the reading thread per MiB)
        evu = run_json([os.path.join(ROOT, "tools", "endpoint_stream"), "1024", str(MIB), "0", "0", "2"], 120, env)
        out["value_endpoint_vtable_unchecked"] = evu.get("GiBps")
        # the same stream with both pairs in latency mode: commands through the resident engine, the receive arena in
        # pinned host memory (no launch chain and no device-to-host copy per call); first hardware run of this
        # combination, in the helper process like the leg above
        evl = run_json([os.path.join(ROOT, "tools", "endpoint_stream"), "256", str(MIB), "1", "1"], 120, env)
        out["value_endpoint_vtable_latency_mode"] = evl.get("GiBps")
        if evl.get("GiBps") is None:
            out["endpoint_vtable_latency_mode"] = evl
        # unary 64 B round trips through the same vtable (tools/endpoint_pingpong.cc): the blocking C ABI, both pairs on
        # the resident engine, and the engine with armed reads (each in a helper process; modes 1-2 are first hardware runs)
        pp = os.path.join(ROOT, "tools", "endpoint_pingpong")
        vt = {}
        # (engine_standing_read: a read stays outstanding on each endpoint, as chttp2 keeps it, and a watcher workgroup of
        #  the engine completes it when the bytes land -- THE vtable latency figure; engine: every read a command, round
        #  4's configuration)
        for mode, name, n in ((0, "launch_chain", 2000), (1, "engine", 20000), (2, "engine_standing_read", 50000)):
            r = run_json([pp, str(n), "64", str(mode)], 90, env)
            vt[name] = {k: r.get(k) for k in ("p50_us", "p95_us", "p99_us", "iters", "watch_hits")} if "p50_us" in r else r
        out["rtt_endpoint_vtable_us"] = vt
    if small is not None:
        sm_steps = max(2, args.steps // 2)
        out["value_ring4096"] = round(wl_leg.user_bytes * sm_steps * world / small["elapsed"] / (1 << 30), 3)
        out["rounds_per_step_ring4096"] = small["rounds"]
        out["value_ring4096_sequential"] = round(wl_leg.user_bytes * sm_steps * world / small_seq["elapsed"] / (1 << 30), 3)
        out["config"]["ring4096_leg"] = ("the headline's connection with a 4 MiB ring (max_sge %d, one Send per round = a ring's worth): paired "
                                         "schedule with the promised credit, the wire inside the planner pair's launch (%d wire workgroups), "
                                         "%d rounds; _sequential: five launches per round, what this leg ran through round 5" % (
                                             args.max_sge, small["wire_groups"], small["rounds"]))
    if args.conns > 1:
        # BASELINE.json configs[3] shape: many connections per GPU, 64 KiB messages, reference-default
        # 4 MiB rings; one op per connection in every launch
        per = args.conn_msgs or max(8, 2048 // args.conns)
        mc = measure(4096, max(2, args.steps // 2), 1, not args.no_verify, False, n_links=args.conns,
                     msgs_per_link=per, payload=64 * 1024)
        out["value_conns%d_64KiB_ring4096" % args.conns] = round(
            mc["user_bytes"] * max(2, args.steps // 2) * world / mc["elapsed"] / (1 << 30), 3)
        out["config"]["multi_connection_leg"] = "%d connections x %d x 64 KiB messages per step, 4 MiB rings, %d rounds" % (
            args.conns, per, mc["rounds"])
        try:  # (the same on the paired schedule: three launches per round instead of five, the credit a round late)
            mp = measure(4096, max(2, args.steps // 2), 1, not args.no_verify, False, n_links=args.conns,
                         msgs_per_link=per, payload=64 * 1024, pipeline=True, sends=1)
            out["value_conns%d_64KiB_ring4096_paired" % args.conns] = round(
                mp["user_bytes"] * max(2, args.steps // 2) * world / mp["elapsed"] / (1 << 30), 3)
        except Exception as e:
            out["conns%d_64KiB_paired_error" % args.conns] = str(e)[:200]
        # ... and as BASELINE.json states it -- BIDIRECTIONAL streaming: the same %d pairs with a link in each direction,
        # every end sender and receiver at once, both directions of every pair in every launch, paired schedule
        try:
            bd_steps = max(2, args.steps // 2)
            mb = measure(4096, bd_steps, 1, not args.no_verify, False, n_links=2 * args.conns, msgs_per_link=per,
                         payload=64 * 1024, pipeline=True, sends=1, bidi=True)
            out["value_conns%d_64KiB_bidi" % args.conns] = round(mb["user_bytes"] * bd_steps * world / mb["elapsed"] / (1 << 30), 3)
            out["conns%d_64KiB_bidi_verified" % args.conns] = mb["verified"]
            out["config"]["multi_connection_bidi_leg"] = (
                "%d pairs x 2 directions x %d x 64 KiB messages per step (both directions counted), 4 MiB rings, paired "
                "schedule, %d rounds" % (args.conns, per, mb["rounds"]))
        except Exception as e:
            out["conns%d_64KiB_bidi_error" % args.conns] = str(e)[:200]
        # ... and in the STEADY STATE: the legs above are a ring's worth per link and step -- three rounds that begin at an
        # empty ring.  512 messages per link (32 MiB through a 4 MiB ring) is what a stream looks like: every round is cut
        # by the credit; the paired schedule sees it a round late, or -- promised credit (round 6: 64 links x (3 + 3)
        # planner workgroups no longer have to be resident at once) -- in the launch that plans the drain.
        try:
            st_steps = max(2, args.steps // 4)
            for key, pr in (("steady", True), ("steady_credit_a_round_late", False)):
                ms_ = measure(4096, st_steps, 1, not args.no_verify, False, n_links=2 * args.conns, msgs_per_link=8 * per,
                              payload=64 * 1024, pipeline=True, sends=1, bidi=True, promise=pr)
                out["value_conns%d_64KiB_bidi_%s" % (args.conns, key)] = round(
                    ms_["user_bytes"] * st_steps * world / ms_["elapsed"] / (1 << 30), 3)
                out["rounds_per_step_conns%d_64KiB_bidi_%s" % (args.conns, key)] = ms_["rounds"]
            out["config"]["multi_connection_bidi_steady_leg"] = (
                "%d pairs x 2 directions x %d x 64 KiB messages per step (both directions counted), 4 MiB rings, paired schedule "
                "with the promised credit / with the credit a round late" % (args.conns, 8 * per))
        except Exception as e:
            out["conns%d_64KiB_bidi_steady_error" % args.conns] = str(e)[:200]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl, ring, min(args.max_sge, 4095))
        if not args.no_extra_legs:
            out["cpu_baseline_ring4096_sge30"] = cpu_baseline(wl, 4096 * 1024, 30, target_s=6.0)
            # ... and the same codec on min(nproc, 8) cores, one connection per core: a CPU's answer to more connections is
            # more cores, and the comparison with one GPU should say what eight of them do
            nthr = max(1, min(os.cpu_count() or 1, 8))
            out["cpu_baseline_ring4096_sge30_%dcores" % nthr] = cpu_baseline(wl, 4096 * 1024, 30, target_s=5.0, threads=nthr)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0 and world == 1 and not args.no_tcp_baseline:
        out["tcp_baseline"] = tcp_baseline()
    if rank == 0:
        print(json.dumps(out))
    grp.close()


if __name__ == "__main__":
    main()
