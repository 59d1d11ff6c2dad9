import sys, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, grpc_rdma_amd as g
from grpc_rdma_amd import stream as gs
g.init(0)
ring=4<<20
tx,rx=g.Pair(ring,4095,0),g.Pair(ring,4095,0); g.connect_pairs(tx,rx)
wl=bench.Workload(g,16)
dst_cap=wl.N+16*(len(wl.lens)*2+64)+4096
dst=g.DeviceBuffer(nbytes=dst_cap)
job=gs.StreamJob(tx,rx,wl.sge,dst.ptr,dst_cap,len(wl.lens)*2+64,3)
for it in range(3):
    r=job.run(gs.RUN_EAGER)
# read the ctl block results: the job's ctl is pinned host memory; expose via debug fn
lib=g.load()
lib.grdma_stream_job_debug.argtypes=[C.c_void_p,C.POINTER(C.c_uint64),C.POINTER(C.c_uint64)]
t=(C.c_uint64*16)(); rr=(C.c_uint64*16)()
lib.grdma_stream_job_debug(job.h,t,rr)
t=[int(x) for x in t]; rr=[int(x) for x in rr]
print("tx stamps (memtime ticks, 100MHz => 10ns):", [t[i]-t[0] for i in range(7)], "m=",t[7])
print("rx: total", rr[1]-rr[0], "rounds", rr[2], "fast", rr[3], "scalar", rr[4], "bulk_took", rr[5], "P,H,V,pe0,pe1,pe[V],n[V],key,vmax,head", rr[6:16])
