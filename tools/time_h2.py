import sys, time, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grpc_rdma_amd as g
from grpc_rdma_amd import h2dev, h2
g.init(0)
lib=g.load()
lib.grdma_h2_last_kernel_us.restype=C.c_double
kus=lambda: lib.grdma_h2_last_kernel_us()
nm=64; M=1048580
buf=g.DeviceBuffer(nbytes=nm*M)
cap=nm*140
sl=g.DeviceBuffer(nbytes=16*cap); hdr=g.DeviceBuffer(nbytes=32*cap)
msgs=[(buf.ptr+i*M, M, 1, 0) for i in range(nm)]
for it in range(3):
    t0=time.perf_counter(); n,w=h2dev.frame_messages(msgs,16384,sl.ptr,cap,hdr.ptr,32*cap); t1=time.perf_counter()
    print("frame: %d slices, %.1f us (incl. alloc/upload/sync), kernel %.1f us"%(n,(t1-t0)*1e6, kus()))
# deframe: build arena with slices from host layout
lay=h2.frame_message(M,1,16384)
one=b"".join(i[1] if i[0]=='inl' else bytes(i[1][1]) for i in lay)
lens=[len(i[1]) if i[0]=='inl' else i[1][1] for i in lay]
arena=bytearray(); table=[]
for m in range(8):
    off0=0
    for L in lens:
        table.append((len(arena),L)); arena+=one[off0:off0+L]+bytes((-L)%16); off0+=L
ab=g.DeviceBuffer(data=bytes(arena)+bytes(64))
p=h2dev.Parser(False)
assert p.open_streams([1]) == 0   # (a client parser: the call was started on stream 1)
for it in range(3):
    t0=time.perf_counter(); err,ev=p.deframe(ab.ptr,table,cap=len(table)*4+64); t1=time.perf_counter()
    print("deframe: %d slices -> %d events err=%d, %.1f us, kernel %.1f us"%(len(table),len(ev),err,(t1-t0)*1e6, kus()))
