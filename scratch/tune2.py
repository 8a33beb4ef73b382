import sys, os, time; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
n,d,dtype,k = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
c=H.gauss(1234,n,d); q=H.gauss(5678,16,d)
dev=torch.device('cuda:0'); tq=torch.from_numpy(q).to(dev)
ix=FlatIPIndex.from_array(c,dtype=dtype)
bytes_=n*d*(2 if dtype=='f16' else 4)
K=2000
outs=[(torch.empty((1,k),dtype=torch.float32,device=dev),torch.empty((1,k),dtype=torch.int64,device=dev)) for _ in range(4)]
def run(streams):
    for i in range(50):
        with torch.cuda.stream(streams[i%len(streams)]): ix.search_device(tq[:1],k,*outs[i%4],asynchronous=True)
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for i in range(K):
        st=streams[i%len(streams)]
        ix.search_device(tq[:1],k,*outs[i%4],asynchronous=True,stream=st)
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/K
s0=torch.cuda.current_stream(); s1=torch.cuda.Stream(); s2=torch.cuda.Stream(); s3=torch.cuda.Stream()
for name,sts in (("1 stream",[s0]),("2 streams",[s1,s2]),("3 streams",[s1,s2,s3])):
    dt=run(sts); print(f"{name}: step={dt*1e6:.1f}us  QPS={1/dt:.0f} effBW={bytes_/dt/1e12:.2f}TB/s",flush=True)
# host-only overhead: time of the python call path with nothing to wait for
o16=(torch.empty((16,k),dtype=torch.float32,device=dev),torch.empty((16,k),dtype=torch.int64,device=dev))
for ov in (0,1):
    ix.debug_option(3,ov)
    for _ in range(5): ix.search_device(tq,k,*o16,asynchronous=True)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(200): ix.search_device(tq,k,*o16,asynchronous=True)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/200/16
    print(f"nq=16 per call overlap={ov}: per-query={dt*1e6:.1f}us QPS={1/dt:.0f} effBW={bytes_/dt/1e12:.2f}TB/s",flush=True)
