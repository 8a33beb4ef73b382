import sys,time; sys.path.insert(0,"/root/repo")
import numpy as np, torch
from lean_explore_amd.index import FlatIPIndex
# config-4-like per-GPU timing: d=768 f16 nq=256 k=100, N rows generated on the device
n=int(sys.argv[1]) if len(sys.argv)>1 else 2_000_000
d,nq,k=768,256,100
g=torch.Generator(device="cuda"); g.manual_seed(1)
c=torch.randn((n,d),device="cuda",generator=g); c/=c.norm(dim=1,keepdim=True)
ix=FlatIPIndex.from_device_tensor(c,dtype="f16"); del c; torch.cuda.empty_cache()
q=torch.randn((nq,d),device="cuda",generator=g); q/=q.norm(dim=1,keepdim=True)
for _ in range(3): ix.search_device(q,k,asynchronous=True)
ix.check(); t0=time.perf_counter()
for _ in range(10): ix.search_device(q,k,asynchronous=True)
ix.check(); dt=(time.perf_counter()-t0)/10
fl=2.0*nq*n*d
print(f"C4-like N={n}: {dt*1e3:.2f} ms/batch  {nq/dt:.0f} QPS  {fl/dt/1e12:.0f} TFLOP/s  {n*d*2/dt/1e12:.2f} TB/s  fallbacks={ix.debug_counter(8)}")
