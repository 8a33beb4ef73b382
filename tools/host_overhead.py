"""Host-side cost of one pipelined batch-1 step (what bounds the step when a shard is small):
python wrapper vs raw ctypes call vs the C side. N=10k so the GPU is never the bottleneck."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from lean_explore_amd import native
from lean_explore_amd.index import FlatIPIndex

n, d, k = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000, 384, 50
rng = np.random.default_rng(0)
c = rng.standard_normal((n, d), dtype=np.float32)
ix = FlatIPIndex.from_array(c)
q = torch.from_numpy(rng.standard_normal((1, d), dtype=np.float32)).cuda()
outs = [(torch.empty((1, k), device="cuda"), torch.empty((1, k), dtype=torch.int64, device="cuda")) for _ in range(16)]
R = 20000
def loop(fn):
    for i in range(200): fn(i)
    ix.check(); torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(R): fn(i)
    th = time.perf_counter() - t          # host enqueue time only
    ix.check(); torch.cuda.synchronize()
    return th / R * 1e6, (time.perf_counter() - t) / R * 1e6
print("wrapper  host/step %.2f us, wall/step %.2f us" % loop(lambda i: ix.search_device(q, k, outs[i & 15][0], outs[i & 15][1], pipeline=True)))
lib, h = native.load(), ix._ensure_built()
s = torch.cuda.current_stream().cuda_stream
ptrs = [(o[0].data_ptr(), o[1].data_ptr()) for o in outs]
qp = q.data_ptr()
fl = native.LS_FLAG_PIPELINE
f = lib.ls_search_device
print("raw ctypes host/step %.2f us, wall/step %.2f us" % loop(lambda i: f(h, qp, 1, k, fl, ptrs[i & 15][0], ptrs[i & 15][1], s)))
