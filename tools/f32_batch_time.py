"""fp32 index, batches of queries: exact f32 MFMA path vs the 8-query scan groups (debug option 4)."""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
for (n, d, k) in [(200000, 1024, 1000), (200000, 384, 100)]:
    c = H.gauss(1234, n, d)
    ix = FlatIPIndex.from_array(c, dtype="f32")
    for nq in (24, 32, 48, 64, 256):
        tq = torch.from_numpy(H.gauss(5678, nq, d)).cuda()
        row = []
        for opt in (1, 0):
            ix.debug_option(4, opt)
            for _ in range(3): ix.search_device(tq, k, asynchronous=True)
            ix.check(); t0 = time.perf_counter()
            R = 20 if nq <= 256 else 5
            for _ in range(R): ix.search_device(tq, k, asynchronous=True)
            ix.check(); row.append((time.perf_counter() - t0) / R)
        fl = 2.0 * nq * n * d
        print(f"N={n} d={d} k={k} nq={nq}: mfma32 {row[0]*1e6:9.1f} us ({fl/row[0]/1e12:6.1f} TFLOP/s)   scan groups {row[1]*1e6:9.1f} us   x{row[1]/row[0]:.2f}  repairs={ix.debug_counter(8)}", flush=True)
    ix.close()
