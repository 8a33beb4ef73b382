"""Small shards: where does a pipelined batch-1 step go? Runs the scan with the selection
piggy-backed (default) and with it as its own launch (debug option 3 = 0); run under
rocprofv3 --kernel-trace --stats to see the two kernels apart."""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
c = H.gauss(1234, n, 384); q = torch.from_numpy(H.gauss(5678, 1, 384)).cuda()
ix = FlatIPIndex.from_array(c, dtype="f32")
for opt in (1, 0):
    ix.debug_option(3, opt)
    for _ in range(200): ix.search_device(q, 50, pipeline=True)
    ix.check(); t0 = time.perf_counter()
    for _ in range(3000): ix.search_device(q, 50, pipeline=True)
    ix.check(); dt = (time.perf_counter() - t0) / 3000
    print(f"N={n} piggyback={opt}: {dt*1e6:.2f} us/step", flush=True)
