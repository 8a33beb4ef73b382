import sys, os, time; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
n,d,dtype,k = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
c=H.gauss(1234,n,d); q=H.gauss(5678,16,d)
dev=torch.device('cuda:0'); tq=torch.from_numpy(q).to(dev)
ix=FlatIPIndex.from_array(c,dtype=dtype)
bytes_=n*d*(2 if dtype=='f16' else 4)
K=3000
outs=[(torch.empty((1,k),dtype=torch.float32,device=dev),torch.empty((1,k),dtype=torch.int64,device=dev)) for _ in range(4)]
for mode in ("async","pipeline"):
    kw=dict(asynchronous=True) if mode=="async" else dict(pipeline=True)
    for i in range(100): ix.search_device(tq[:1],k,*outs[i%4],**kw)
    ix.check()
    t0=time.perf_counter()
    for i in range(K): ix.search_device(tq[:1],k,*outs[i%4],**kw)
    th=time.perf_counter()-t0
    ix.check(); dt=(time.perf_counter()-t0)/K
    print(f"{mode}: step={dt*1e6:.1f}us host_issue={th/K*1e6:.1f}us QPS={1/dt:.0f} effBW={bytes_/dt/1e12:.2f}TB/s ({bytes_/dt/8e12*100:.1f}% of 8TB/s)",flush=True)
