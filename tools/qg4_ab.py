import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from lean_explore_amd.index import FlatIPIndex
from oracle import oracle
from tests import helpers as H
n, d, nq, k = 200_000, 384, 1024, 100
c = H.gauss(1234, n, d); q = H.gauss(5678, nq, d)
ix = FlatIPIndex.from_array(c, dtype="f16")
tq = torch.from_numpy(q).cuda()
res = {}
for shape in (0, 1, 0, 1):
    ix.debug_option(18, shape)
    s, i = ix.search_device(tq, k, asynchronous=True); ix.check()
    rep0 = ix.debug_counter(8)
    for _ in range(5): ix.search_device(tq, k, pipeline=True)
    ix.check()
    ix.set_profiling(True)
    for _ in range(50): ix.search_device(tq, k, asynchronous=True)
    ix.check(); ms, _ = ix.last_kernel_ms(); ix.set_profiling(False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): ix.search_device(tq, k, pipeline=True)
    ix.check(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
    res[shape] = (s.cpu().numpy(), i.cpu().numpy())
    print(f"shape {'rs2/qg4' if shape else 'default'}: pass kernel {ms*1e3:.1f} us, pipelined batch {dt*1e6:.1f} us, repaired so far {ix.debug_counter(8)}", flush=True)
print("identical results:", np.array_equal(res[0][0], res[1][0]), np.array_equal(res[0][1], res[1][1]))
Dr, Ir = oracle.c_search(c, q[:16], k, f16=True); _, _, S = oracle.np_search(c, q[:16], k, f16=True)
print(oracle.compare_topk(res[1][0][:16], res[1][1][:16], Dr, Ir, S))
