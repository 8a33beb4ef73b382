"""Synchronous host API (ls_search: what the reference's index.search call maps to, search/engine.py:250),
PCIe + sync inclusive, with the selection inside the scan launch (debug option 9 = 1, default) and as a
separate launch (0), interleaved on one box:  gpurun -- 'python tools/hostapi_time.py'"""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H

def lat(ix, q, k, calls):
    for _ in range(30): ix.search(q, k, normalize=True)
    t = np.empty(calls)
    for i in range(calls):
        t0 = time.perf_counter(); ix.search(q, k, normalize=True); t[i] = time.perf_counter() - t0
    return np.median(t) * 1e6, t.mean() * 1e6

for (n, d, dt, nq, k) in [(200000, 384, 'f32', 1, 50), (200000, 1024, 'f32', 1, 1000), (25000, 384, 'f32', 1, 50),
                          (200000, 384, 'f32', 8, 50), (200000, 384, 'f16', 1, 50)]:
    c = H.gauss(1234, n, d); q = H.gauss(5678, nq, d)
    ix = FlatIPIndex.from_array(c, dtype=dt)
    res = {0: [], 1: []}
    for rep in range(3):
        for mode in (1, 0):
            ix.debug_option(9, mode)
            res[mode].append(lat(ix, q, k, 300))
    f = lambda v: "/".join(f"{a:.1f}" for a, _ in v)
    print(f"N={n} d={d} {dt} nq={nq} k={k}: same-launch selection p50 {f(res[1])} us | separate launch p50 {f(res[0])} us", flush=True)
    ix.close()
