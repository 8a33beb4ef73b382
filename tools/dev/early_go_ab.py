import sys; sys.path.insert(0, '.')
import numpy as np
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
from tools.concurrent_callers_lib import run
c = H.gauss(1234, 200_000, 384); q = H.gauss(5678, 16, 384)
ix = FlatIPIndex.from_array(c)
for _ in range(50): ix.search(q[:1], 50, normalize=True)
for rep in range(3):
    for cap in (8, 2):
        ix.debug_option(21, cap)
        row = []
        for T in (2, 4, 8, 16):
            qps, p50 = run(ix, q, 50, T)
            row.append(f"{T}: {qps:7.0f} q/s p50 {p50:5.1f}")
        print(f"python threads, option 21 = {cap}: " + " | ".join(row), flush=True)
ix.close()
