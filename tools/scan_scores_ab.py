"""Pipelined single-query device calls (the headline's loop) with / without the score vector (debug option 19 = 1 / 2),
alternating on one handle, wall clock per call:  gpurun -- 'python tools/scan_scores_ab.py'"""
import sys, time; sys.path.insert(0, '/root/repo')
import torch
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H

for n, d, k in ((200_000, 384, 50), (200_000, 1024, 1000), (1_000_000, 384, 50)):
    ix = FlatIPIndex.from_array(H.gauss(1234, n, d)); q = torch.from_numpy(H.gauss(5678, 1, d)).cuda()
    o = (torch.empty((1, k), dtype=torch.float32, device='cuda'), torch.empty((1, k), dtype=torch.int64, device='cuda'))
    row = []
    for rep in range(3):
        for opt in (1, 2):
            ix.debug_option(19, opt)
            for _ in range(200): ix.search_device(q, k, o[0], o[1], pipeline=True)
            ix.check(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3000): ix.search_device(q, k, o[0], o[1], pipeline=True)
            ix.check(); torch.cuda.synchronize()
            row.append(f"{'no S' if opt == 1 else 'with S'} {(time.perf_counter() - t0) / 3000 * 1e6:6.2f}")
    print(f"N={n} d={d} k={k}: " + " | ".join(row) + f" us/call | served again: {ix.debug_counter(25)}", flush=True)
    ix.close()
