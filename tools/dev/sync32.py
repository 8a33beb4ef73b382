import sys, time; sys.path.insert(0, __import__('os').path.abspath(__import__('os').path.join(__import__('os').path.dirname(__file__), '..', '..')))
import numpy as np
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
for d, k in ((384, 50), (1024, 1000)):
    ix = FlatIPIndex.from_array(H.gauss(1234, 200_000, d))
    for nq in (1, 8, 16, 17, 24, 32):
        q = H.gauss(5, nq, d)
        for _ in range(30): ix.search(q, k)
        lat = []
        for _ in range(200):
            t0 = time.perf_counter(); ix.search(q, k); lat.append(time.perf_counter() - t0)
        print(f"sync ls_search d={d} k={k} nq={nq}: p50 {np.median(lat)*1e6:.1f} us; launches/call {ix.debug_counter(11)}; retries {ix.debug_counter(20)}, reserved {ix.debug_counter(25)}", flush=True)
    ix.close()
