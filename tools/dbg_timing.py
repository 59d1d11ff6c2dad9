import sys, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, grpc_rdma_amd as g
from grpc_rdma_amd import stream as gs
g.init(0)
ring=int(os.environ.get('RING_KB','4096'))<<10
tx,rx=g.Pair(ring,4095,0),g.Pair(ring,4095,0); g.connect_pairs(tx,rx)
wl=bench.Workload(g,int(os.environ.get('MSGS','16')))
dst_cap=wl.N+16*(len(wl.lens)*2+64)+4096
dst=g.DeviceBuffer(nbytes=dst_cap)
job=gs.StreamJob(tx,rx,wl.sge,dst.ptr,dst_cap,len(wl.lens)*2+64,int(os.environ.get('ROUNDS','3')))
for it in range(3):
    r=job.run(gs.RUN_EAGER)
# read the ctl block results: the job's ctl is pinned host memory; expose via debug fn
lib=g.load()
lib.grdma_stream_job_debug.argtypes=[C.c_void_p,C.POINTER(C.c_uint64),C.POINTER(C.c_uint64)]
t=(C.c_uint64*16)(); rr=(C.c_uint64*16)()
lib.grdma_stream_job_debug(job.h,t,rr)
t=[int(x) for x in t]; rr=[int(x) for x in rr]
print("tx stamps (memtime ticks, 100MHz => 10ns):", [t[i]-t[0] for i in range(7)], "m=",t[7], "loaded", t[8]-t[0])
print("rx: total", rr[1]-rr[0], "rounds", rr[2], "fast", rr[3], "scalar", rr[4], "bulk_took", rr[5], "P", rr[6], "V", rr[7], "cycles: period", rr[8], "probe", rr[9], "pass0", rr[10], "pass1", rr[11], "pass2", rr[12], "loop_total", rr[13], "prologue", rr[14], "period+predict", rr[15])

hist=(C.c_uint32*1024)(); cnt=C.c_uint64(0); per=C.c_uint32(0)
lib.grdma_pair_debug_hist.argtypes=[C.c_void_p,C.POINTER(C.c_uint32),C.POINTER(C.c_uint64),C.POINTER(C.c_uint32)]
lib.grdma_pair_debug_hist(rx.h,hist,C.byref(cnt),C.byref(per))
n=cnt.value; H=min(n,1024)
seq=[int(hist[(n-H+i)%1024]) for i in range(H)]
print("hist count",n,"period",per.value)
# run-length print of the newest 400 entries
tail=seq[-400:]
out=[]; i=0
while i<len(tail):
    if i+1<len(tail) and (tail[i],tail[i+1])==(32,16400):
        k=0
        while i+1<len(tail) and (tail[i],tail[i+1])==(32,16400): k+=1; i+=2
        out.append("(32,16400)x%d"%k)
    else:
        out.append(str(tail[i])); i+=1
print(" ".join(out))
