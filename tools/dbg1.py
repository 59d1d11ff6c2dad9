import sys, random
sys.path.insert(0, '.')
import grpc_rdma_amd as g
from oracle import pyorc
g.init(0)
import os
seed=int(os.environ.get("SEED","0")); flags=int(os.environ.get("FLAGS","0"))
rng = random.Random(1000 + seed)
R = rng.choice([64, 256, 4096, 65536]); sge = rng.choice([1, 3, 30, 200])
print("R",R,"sge",sge)
a, b = g.Pair(R, sge, flags), g.Pair(R, sge, flags); g.connect_pairs(a,b)
o = pyorc.OracleLink(R, sge)
sizes = [1, 2, 7, 8, 9, 15, 16, 17, 23, 24, 100, 255, 256, 257, R // 3, R]
for step in range(40):
    op = rng.random()
    if op < 0.5:
        n = rng.randint(1, 8)
        sl = [bytes(rng.getrandbits(8) for _ in range(rng.choice(sizes))) for _ in range(n)]
        bi = rng.randrange(len(sl[0])) if rng.random() < 0.3 else 0
        bufs = [g.DeviceBuffer(data=s, offset=rng.randrange(16)) for s in sl]
        s_g = a.Send(bufs, bi); s_o = o.send(0, sl, bi)
        print(step, "send lens", [len(s) for s in sl], "bi", bi, "->", s_g, s_o, a.last_wrs(), o.last_wrs(0))
    elif op < 0.75:
        cap = rng.choice([1, 3, 8, 64, 256, R])
        x=b.Recv(cap); y=o.recv(1,cap)
        print(step, "recv cap", cap, len(x), len(y), x==y)
    else:
        got,_=b.endpoint_read(1); exp,_al=o.endpoint_read(1)
        print(step, "epread", [len(x) for x in got], len(exp), (got[0] if got else b"")==exp, "alloc", _al)
    x=b.ring_mem(); y=o.ring_mem(1)
    d=[i for i in range(R) if x[i]!=y[i]]
    print("   state g", b.state()); print("   state o", o.state(1))
    if d:
        print("   DIFF at", d[:20], "count", len(d))
        i=d[0]&~15
        print("   g:", x[i-16:i+48].hex()); print("   o:", y[i-16:i+48].hex())
        break
