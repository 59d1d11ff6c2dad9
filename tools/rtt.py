import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grpc_rdma_amd as g
from grpc_rdma_amd import h2
g.init(0)
msg = bytes([0x0A, 64]) + bytes(range(64))
items = h2.frame_message(len(msg), 1)
slices = [i[1] if i[0] == "inl" else msg[i[1][0]:i[1][0] + i[1][1]] for i in items]
import ctypes as C
lib=g.load()
for mode in ('engine',):
    a, b = g.Pair(4 << 20, 30), g.Pair(4 << 20, 30); g.connect_pairs(a, b)
    a.set_latency_mode(bool(mode)); b.set_latency_mode(bool(mode))
    if mode == 'engine': assert lib.grdma_engine_start() == 0, lib.grdma_last_error()
    n = 2000 if mode else 300
    try:
        rtt, ph = g.pingpong(a, b, slices, slices, iters=n, warmup=100)
    except Exception as e:
        print('FAILED', e); sys.stdout.flush(); os._exit(1)
    rtt.sort()
    print("latency_mode=%s p50 %.1f us p95 %.1f p99 %.1f min %.1f | phases us: %s" % (
        mode, rtt[n // 2] / 1e3, rtt[int(n * .95)] / 1e3, rtt[int(n * .99)] / 1e3, rtt[0] / 1e3,
        [round(x / n / 1e3, 1) for x in ph]))
    if mode == 'engine':
        d=(C.c_uint64*5)(); lib.grdma_engine_debug(d); ops=2*(n+100)
        print('engine cycles per op: send load %d body %d | drain load %d body %d' % (d[0]//ops, d[1]//ops, d[2]//ops, d[3]//ops))
        tk=(C.c_uint64*8)(); lib.grdma_tx_small_ticks(tk); cnt=max(1,int(tk[6]))
        print('small-send phases (cycles per send): loads %d pricing %d copies-issue %d copies-ack %d bookkeeping %d release %d' % tuple(int(tk[i])//cnt for i in range(6)))
        lib.grdma_engine_stop()
        lib.grdma_express_drains.restype=C.c_uint64
        print('express drains:', lib.grdma_express_drains(), 'of', ops, 'drains')
        td=(C.c_uint64*16)(); rd=(C.c_uint64*16)(); lib.grdma_pair_last_dbg.argtypes=[C.c_void_p,C.POINTER(C.c_uint64),C.POINTER(C.c_uint64)]
        lib.grdma_pair_last_dbg(a.h,td,rd)
        print('tx stamps', [int(td[i])-int(td[0]) for i in range(10)])
        print('rx: loop_end %d before_results %d end %d (from begin)' % (int(rd[13])-int(rd[0]), int(rd[14])-int(rd[0]), int(rd[1])-int(rd[0])), 'rounds', int(rd[2]), 'fast', int(rd[3]), 'scalar', int(rd[4]))
