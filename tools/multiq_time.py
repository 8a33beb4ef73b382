import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
n,d,dtype,k = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
c=H.gauss(1234,n,d); q=H.gauss(5678,64,d)
ix=FlatIPIndex.from_array(c,dtype=dtype); tq=torch.from_numpy(q).cuda()
ix.debug_option(4,0)   # keep everything on the scan path
bytes_=n*d*(2 if dtype=='f16' else 4)
for mq in (0,1):
    ix.debug_option(6,mq)
    for nq in (1,2,4,8,16,64):
        for _ in range(3): ix.search_device(tq[:nq],k,asynchronous=True)
        torch.cuda.synchronize(); t0=time.perf_counter(); K=20
        for _ in range(K): ix.search_device(tq[:nq],k,asynchronous=True)
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/K
        print(f"multi_query={mq} nq={nq:3d}: {dt*1e6:8.1f} us/call  {dt/nq*1e6:7.1f} us/query  {nq/dt:9.0f} QPS", flush=True)
