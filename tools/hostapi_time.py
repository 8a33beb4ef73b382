import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
for (n,d,dt,nq,k) in [(200000,384,'f32',1,50),(200000,1024,'f32',1,1000),(200000,384,'f16',1024,100)]:
    c=H.gauss(1234,n,d); q=H.gauss(5678,nq,d)
    ix=FlatIPIndex.from_array(c,dtype=dt)
    for _ in range(20): ix.search(q,k)
    t0=time.perf_counter(); K=300 if nq==1 else 30
    for _ in range(K): ix.search(q,k)
    t=(time.perf_counter()-t0)/K
    print(f"host API (PCIe + sync inclusive) N={n} d={d} {dt} nq={nq} k={k}: {t*1e6:.1f} us/call  {nq/t:.0f} QPS", flush=True)
    ix.close()
