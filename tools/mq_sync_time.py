import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
for d, k in ((384, 50), (1024, 50), (1024, 1000)):
    c = H.gauss(1234, 200_000, d); ix = FlatIPIndex.from_array(c)
    for nq in (1, 2, 8, 16):
        q = H.gauss(5678, nq, d)
        row = []
        for same in (1, 0):
            ix.debug_option(9, same)
            for _ in range(30): ix.search(q, k, normalize=True)
            lat = []
            for _ in range(200):
                t0 = time.perf_counter(); ix.search(q, k, normalize=True); lat.append(time.perf_counter() - t0)
            row.append(f"{'same-launch' if same else 'own launch'} p50 {np.median(lat)*1e6:.1f} us")
        ix.debug_option(9, 1)
        print(f"N=200000 d={d} k={k} nq={nq} synchronous ls_search: " + " | ".join(row), flush=True)
    ix.close()
