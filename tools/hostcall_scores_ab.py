"""Synchronous single-query host calls (the reference's call shape) with / without score vectors (debug option 19),
alternating on one handle:  gpurun -- 'python tools/hostcall_scores_ab.py'"""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H

for n, d, k in ((200_000, 384, 50), (200_000, 1024, 1000)):
    ix = FlatIPIndex.from_array(H.gauss(1234, n, d)); q = H.gauss(5678, 64, d)
    for _ in range(100): ix.search(q[:1], k, normalize=True)
    row = []
    for rep in range(3):
        for opt in (1, 0):
            ix.debug_option(19, opt)
            lat = []
            for i in range(600):
                t0 = time.perf_counter(); ix.search(q[i % 64:i % 64 + 1], k, normalize=True); lat.append(time.perf_counter() - t0)
            row.append(f"{'no S' if opt else 'with S'} p50 {np.median(lat) * 1e6:6.1f}")
    print(f"N={n} d={d} k={k}: " + " | ".join(row) + f" | served again: {ix.debug_counter(25)}", flush=True)
    ix.close()
