"""Config 2 step time vs the number of scan workgroups per launch (debug option 7), one process,
interleaved rounds: is the tile count per wave (25000 tiles / (4 * blocks)) what sets the tail?"""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
n, d, k = (int(sys.argv[1]) if len(sys.argv) > 1 else 200000), (int(sys.argv[2]) if len(sys.argv) > 2 else 384), 50
c = H.gauss(1234, n, d); q = torch.from_numpy(H.gauss(5678, 1, d)).cuda()
ix = FlatIPIndex.from_array(c, dtype="f32")
cands = [0, 384, 448, 480, 512, 521, 544, 568, 600, 625, 640, 696, 768, 782, 896, 1024]
res = {b: [] for b in cands}
for rnd in range(3):
    for b in cands:
        ix.debug_option(7, b)
        for _ in range(300): ix.search_device(q, k, pipeline=True)
        ix.check(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3000): ix.search_device(q, k, pipeline=True)
        ix.check(); torch.cuda.synchronize()
        res[b].append((time.perf_counter() - t0) / 3000 * 1e6)
for b in cands:
    print(f"blocks={b:5d}: {np.median(res[b]):7.2f} us/step  {['%.2f' % x for x in res[b]]}")
