#!/usr/bin/env python3
"""Throughput of the persistent link engine at the bench configurations, with the leaders' wait
split (tools; bench.py carries the official legs).  usage: engine_perf.py [config ...]
configs: big (128 MiB ring, sge 4095), r4m (4 MiB ring, sge 4095), ref (4 MiB ring, sge 30),
         conns (32 links x 64 KiB, 4 MiB rings), mixed (sizes 1 B..4 MiB, 4 MiB ring, sge 30)"""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(g, gs, name, ring_kb, max_sge, n_links, msgs, payload, steps=10, flags=0, wls=None):
    ring = ring_kb * 1024
    wls = wls or [bench.Workload(g, msgs, payload) for _ in range(n_links)]
    links, keep = [], []
    for w in wls:
        tx, rx = g.Pair(ring, max_sge, flags), g.Pair(ring, max_sge, flags)
        g.connect_pairs(tx, rx)
        scap = len(w.lens) * 2 + 64 + w.N // 256
        dst_cap = w.N + 32 * scap + 4096
        dst = g.DeviceBuffer(nbytes=dst_cap)
        links.append((tx, rx, w.sge, dst.ptr, dst_cap, scap))
        keep.append((tx, rx, dst))
    job = gs.MultiStreamJob(links, 8)
    r = job.run(gs.RUN_ENGINE)
    assert r.done
    for _ in range(2):
        job.launch_engine()
    job.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        job.launch_engine()
    job.sync()
    dt = time.perf_counter() - t0
    user = sum(w.user_bytes for w in wls)
    st = job.engine_stats(0)
    r = job.run(gs.RUN_ENGINE)
    out = {"config": name, "GiBps": round(user * steps / dt / (1 << 30), 2), "ms_per_step": round(1e3 * dt / steps, 4),
           "ms_single": round(r.ms_total, 4), "sends": st["sends"], "chunks": st["chunks"],
           "entries": [st["gather_entries"], st["wire_entries"], st["scatter_entries"]],
           "waves": [st["gather_waves"], st["wire_waves"], st["scatter_waves"]], "team": st["team"],
           "staging": st["staging_buffers"],
           "ms": {k: round(st[k] / 1e5, 3) for k in ("tx_wait_slots", "tx_wait_credit", "tx_price", "tx_publish", "tx_total",
                                                     "rx_wait_data", "rx_wait_scatter", "rx_walk", "rx_fast", "rx_scalar", "rx_total", "rx_emit",
                                                     "tx_ph_load", "tx_ph_price", "tx_ph_count", "tx_ph_emit")}}
    print(json.dumps(out), flush=True)
    if os.environ.get("ENGINE_TRACE"):
        names = {1: "tx priced", 2: "tx published", 3: "tx released up to", 4: "rx waits for", 5: "rx landed",
                 6: "rx planned", 7: "rx credit after chunk", 8: "rx round complete", 9: "takes entry", 10: "dependency met", 11: "done entry"}
        for t, who, tag, arg in job.engine_trace(0):
            print("  %9.1f us  %s %s %d" % (t, ["TX", "RX", "  g0", "    w0", "      s0"][who], names.get(tag, tag), arg))
    job.close()
    for tx, rx, dst in keep:
        tx.close(); rx.close(); dst.free()
    return out


MixedWorkload = bench.MixedWorkload


def run_burst(g, gs, name, ring_kb, max_sge, msgs, burst, steps=6):
    """graph schedule with `burst` Sends per round"""
    ring = ring_kb * 1024
    w = bench.Workload(g, msgs, bench.MIB)
    tx, rx = g.Pair(ring, max_sge), g.Pair(ring, max_sge)
    g.connect_pairs(tx, rx)
    scap = len(w.lens) * 2 + 64 + w.N // 256
    dst_cap = w.N + 32 * scap + 4096
    dst = g.DeviceBuffer(nbytes=dst_cap)
    job = gs.MultiStreamJob([(tx, rx, w.sge, dst.ptr, dst_cap, scap)], 2 * (len(w.lens) // max(1, max_sge * burst) + 8) + w.E // (ring // 2) * 4)
    job.set_burst(burst)
    r = job.run(gs.RUN_EAGER)
    assert r.done, (r.bytes_delivered, w.N)
    rounds = int(r.rx_rounds) + 1
    job.set_rounds(rounds)
    r = job.run(gs.RUN_GRAPH)
    assert r.done
    for _ in range(2):
        job.launch()
    job.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        job.launch()
    job.sync()
    dt = time.perf_counter() - t0
    inst = job.run(gs.RUN_INSTRUMENTED)
    us = {gs.CLASS_NAMES[i]: round(1e3 * inst.ms_class[i] / max(1, int(inst.launches_class[i])), 1) for i in range(len(gs.CLASS_NAMES))}
    print(json.dumps({"config": name, "burst": burst, "GiBps": round(w.user_bytes * steps / dt / (1 << 30), 2),
                      "ms_per_step": round(1e3 * dt / steps, 3), "rounds": rounds, "us_per_launch": us}), flush=True)
    job.close()
    tx.close(); rx.close(); dst.free()


def main():
    import torch  # noqa: F401  (device context like bench.py)
    import grpc_rdma_amd as g
    from grpc_rdma_amd import stream as gs
    g.init(0)
    which = sys.argv[1:] or ["big", "r4m", "ref", "conns", "mixed"]
    for w in which:
        if w == "big":
            run(g, gs, "ring128m_sge4095", 131072, 4095, 1, 256, bench.MIB)
        elif w == "bigd":
            run(g, gs, "ring128m_sge4095_direct", 131072, 4095, 1, 256, bench.MIB, flags=2)
        elif w == "r4m":
            run(g, gs, "ring4m_sge4095", 4096, 4095, 1, 256, bench.MIB)
        elif w == "ref":
            run(g, gs, "ring4m_sge30", 4096, 30, 1, 256, bench.MIB)
        elif w == "conns":
            run(g, gs, "conns32_64k", 4096, 4095, 32, 64, 64 * 1024)
        elif w == "mixed":
            run(g, gs, "mixed_ring4m_sge30", 4096, 30, 1, 0, 0, wls=[MixedWorkload(g, 64)])
        elif w.startswith("burst"):
            run_burst(g, gs, "ring4m_sge30", 4096, 30, 256, int(w[5:] or 16))
        elif w == "mixedbig":
            run(g, gs, "mixed_ring128m_sge4095", 131072, 4095, 1, 0, 0, wls=[MixedWorkload(g, 64)])


if __name__ == "__main__":
    main()
