import os, sys; sys.path.insert(0, '/root/repo')
import torch
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
for (n, d) in ((200_000, 384), (200_000, 1024)):
    c = H.gauss(1234, n, d)
    ix = FlatIPIndex.from_array(c)
    q = torch.from_numpy(H.gauss(5678, 16, d)).cuda()
    for blocks in tuple(int(x) for x in os.environ.get('MQ_BLOCKS', '0,256,0,482,512').split(',')):
        ix.debug_option(7, blocks)
        for _ in range(30): ix.search_device(q, 50, pipeline=True)
        ix.check(); ix.set_profiling(True)
        for _ in range(200): ix.search_device(q, 50, pipeline=True)
        ix.check(); ms, _ = ix.last_kernel_ms(); ix.set_profiling(False)
        print(f"N={n} d={d} nq=16 blocks={blocks}: {ms*1e3:.1f} us/launch", flush=True)
    ix.close()
