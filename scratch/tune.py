import sys, os, time; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
n,d,dtype,k = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
c=H.gauss(1234,n,d); q=H.gauss(5678,1,d)
dev=torch.device('cuda:0'); tq=torch.from_numpy(q).to(dev)
ix=FlatIPIndex.from_array(c,dtype=dtype)
for _ in range(50): ix.search_device(tq,k,asynchronous=True)
torch.cuda.synchronize()
t0=time.perf_counter(); K=1000
for _ in range(K): ix.search_device(tq,k,asynchronous=True)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/K
ix.set_profiling(True); sm=[]
for _ in range(30):
    ix.search_device(tq,k,asynchronous=True); torch.cuda.synchronize(); sm.append(ix.last_kernel_ms())
sm=np.array(sm).mean(0)
bytes_=n*d*(2 if dtype=='f16' else 4)
print(f"BPC={os.environ.get('LS_SCAN_BPC')} NT={os.environ.get('LS_SCAN_NT')} ALT={os.environ.get('LS_SCAN_ALT')} step={dt*1e6:.1f}us scan={sm[0]*1e3:.1f}us scan+fin={sm[1]*1e3:.1f}us  scanBW={bytes_/sm[0]/1e9:.2f}TB/s stepBW={bytes_/dt/1e12:.2f}TB/s", flush=True)
