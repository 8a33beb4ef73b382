import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
c=H.gauss(1234,200000,384); q=H.gauss(5678,1024,384)
ix=FlatIPIndex.from_array(c,dtype='f16'); tq=torch.from_numpy(q).cuda()
for _ in range(5): ix.search_device(tq,100,asynchronous=True)
torch.cuda.synchronize()
ix.set_profiling(True)
t0=time.perf_counter()
for _ in range(30): ix.search_device(tq,100,asynchronous=True)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/30
print(f"step={dt*1e6:.1f}us gemm_main={ix.last_kernel_ms()[0]*1e3:.1f}us")
