"""Automatic scan workgroup count (last-round fill rule) vs the old fixed 2 per CU, interleaved."""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
for (n, d, dt, k) in [(200000, 384, "f32", 50), (200000, 1024, "f32", 1000), (1000000, 384, "f32", 50), (200000, 384, "f16", 50), (200000, 768, "f32", 100)]:
    c = H.gauss(1234, n, d); q = torch.from_numpy(H.gauss(5678, 1, d)).cuda()
    ix = FlatIPIndex.from_array(c, dtype=dt)
    res = {0: [], 512: []}
    for rnd in range(3):
        for b in (0, 512):
            ix.debug_option(7, b)
            for _ in range(200): ix.search_device(q, k, pipeline=True)
            ix.check(); torch.cuda.synchronize(); t0 = time.perf_counter()
            R = 2000 if n <= 200000 else 600
            for _ in range(R): ix.search_device(q, k, pipeline=True)
            ix.check(); torch.cuda.synchronize()
            res[b].append((time.perf_counter() - t0) / R * 1e6)
    print(f"N={n} d={d} {dt} k={k}: auto {np.median(res[0]):7.2f} us/step   512 blocks {np.median(res[512]):7.2f} us/step", flush=True)
    ix.close()
