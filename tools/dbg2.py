import sys, random, os
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grpc_rdma_amd as g
from oracle import pyorc
g.init(0)
R=1<<18; sizes=[300, 511, 700, 5000, 16384]; seed=5; flags=0
rng = random.Random(seed)
a, b = g.Pair(R, 4095, flags), g.Pair(R, 4095, flags); g.connect_pairs(a,b)
o = pyorc.OracleLink(R, 4095)
KEYS=["head","moving_head","remain","remote_tail","remote_head","internal_read_size","credit_msgs","partial_write"]
for cycle in range(6):
    while True:
        sl = [bytes(rng.getrandbits(8) for _ in range(rng.choice(sizes))) for _ in range(rng.randint(1, 60))]
        bufs = [g.DeviceBuffer(data=s, offset=rng.randrange(16)) for s in sl]
        s_g = a.Send(bufs); s_o = o.send(0, sl)
        assert s_g == s_o
        if s_g < sum(len(x) for x in sl) or rng.random() < 0.15:
            break
    if rng.random() < 0.5:
        cap = rng.choice([1, 5, 100, 256, 300])
        x=b.Recv(cap); y=o.recv(1, cap); print("recv",cap,len(x),x==y)
    limit = rng.choice([4096, 4096, 7, 1])
    print("cycle",cycle,"limit",limit,"state before g", {k:b.state()[k] for k in KEYS}, "o", o.state(1))
    got, wb = b.endpoint_read(limit)
    exp = []
    while len(exp) < limit:
        s, _al = o.endpoint_read(1)
        if not s: break
        exp.append(s)
    print("  slices", len(got), len(exp), [len(x) for x in got]==[len(x) for x in exp], got==exp)
    sa=a.state(); oa=o.state(0); sb=b.state(); ob=o.state(1)
    print("  a:", {k:(sa[k],oa[k]) for k in KEYS if sa[k]!=oa[k]}, " b:", {k:(sb[k],ob[k]) for k in KEYS if sb[k]!=ob[k]})
