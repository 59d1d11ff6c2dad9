import os, sys, random
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))


import grpc_rdma_amd as g
from grpc_rdma_amd import stream as gs
from oracle import pyorc
import test_gpu_stream_job as T
g.init(0)
R, max_sge, sends, n_msgs, msg_len = 1 << 20, 64, 2, 12, 200000
rng = random.Random(R % 73 + sends)
body = bytes(rng.getrandbits(8) for _ in range(min(msg_len, 4096))) * (msg_len // min(msg_len, 4096) + 1)
slices=[]
for i in range(n_msgs):
    wire, lens = pyorc.h2_frame_message(body[:msg_len], stream_id=2*i+1)
    off=0
    for ln in lens:
        slices.append(wire[off:off+ln]); off+=ln
rng2 = random.Random(5)
bufs = [g.DeviceBuffer(data=s, offset=rng2.randrange(16)) for s in slices]
tx, rx = g.Pair(R, max_sge, 0), g.Pair(R, max_sge, 0)
g.connect_pairs(tx, rx)
N = sum(len(s) for s in slices)
dst_cap = N + 32 * (2 * len(slices) + 64) + 4096
dst = g.DeviceBuffer(nbytes=dst_cap)
sge = [(b.ptr, len(s)) for b, s in zip(bufs, slices)]
job = gs.MultiStreamJob([(tx, rx, sge, dst.ptr, dst_cap, 2 * len(slices) + 64)], 40)
job.set_pipeline(True); job.set_sends(sends); job.set_promised_credit(True)
o = pyorc.OracleLink(R, max_sge)
def oracle_pass():
    idx, byte, rounds, out = 0, 0, 0, []
    while idx < len(slices):
        for _k in range(sends):
            if idx >= len(slices): break
            sent = o.send(0, slices[idx:], byte); left = sent
            while left > 0:
                room = len(slices[idx]) - byte
                if left >= room: left -= room; idx += 1; byte = 0
                else: byte += left; left = 0
        rounds += 1
        nread = 0
        while True:
            s, _a = o.endpoint_read(1)
            if not s: break
            out.append(len(s)); nread += len(s)
        print("   oracle round", rounds, "tail", o.state(0)["remote_tail"], "rhead", o.state(0)["remote_head"], "read", nread)
    return out
for p, mode in enumerate([gs.RUN_EAGER, gs.RUN_GRAPH, gs.RUN_GRAPH]):
    if p == 1: job.set_rounds(18)
    c0 = T._fast_counts(g)
    r = job.run(mode)
    c1 = T._fast_counts(g)
    ds = job.delivered_slices(0)
    print("pass", p, "done", r.done, "tx_rounds", r.tx_rounds, "rx_rounds", r.rx_rounds, "counts", [a-b for a,b in zip(c1,c0)], "tx", tx.state()["remote_tail"], tx.state()["remote_head"])
    el = oracle_pass()
    gl = [n for _o, n in ds]
    print("   slices equal:", gl == el, len(gl), len(el))
