"""Host-side cost of the in-library sharded handle: synchronous ls_search latency with G shards rehearsed on
ONE GPU (device_ids = [0] * G: the exchange is copies, the shards' kernels share the device), against the
plain single-device handle:   gpurun -- 'python tools/group_overhead.py'"""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H

def p50(ix, q, k, calls=400):
    for _ in range(50): ix.search(q, k, normalize=True)
    t = np.empty(calls)
    for i in range(calls):
        t0 = time.perf_counter(); ix.search(q, k, normalize=True); t[i] = time.perf_counter() - t0
    return np.median(t) * 1e6

for n, d, k in ((200_000, 384, 50), (200_000, 1024, 1000)):
    c = H.gauss(1234, n, d); q = H.gauss(5678, 1, d)
    ix = FlatIPIndex.from_array(c); base = p50(ix, q, k); ix.close()
    row = [f"plain {base:.1f}"]
    for G in (1, 2, 4, 8):
        ix = FlatIPIndex.from_array(c, devices=[0] * G)
        row.append(f"G={G} {p50(ix, q, k):.1f} (host enqueue {ix.debug_counter(18) / 1e3:.1f})")
        ix.close()
    print(f"N={n} d={d} k={k} nq=1 synchronous call, p50 us: " + " | ".join(row), flush=True)
