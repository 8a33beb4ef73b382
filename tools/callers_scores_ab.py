"""A/B of debug option 19 (ls_mq launches of synchronous host calls write no score vectors) under T concurrent
callers on one handle:  gpurun -- 'python tools/callers_scores_ab.py'"""
import sys, threading, time; sys.path.insert(0, '/root/repo')
import numpy as np
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
from tools.concurrent_callers_lib import run

for n, d, k in ((200_000, 384, 50), (200_000, 1024, 1000), (200_000, 1024, 50)):
    ix = FlatIPIndex.from_array(H.gauss(1234, n, d)); q = H.gauss(5678, 16, d)
    for _ in range(50): ix.search(q[:1], k, normalize=True)
    for T in (4, 8, 16):
        row = []
        for rep in range(2):
            for opt in (1, 0):
                ix.debug_option(19, opt)
                qps, p50 = run(ix, q, k, T, 0.7)
                row.append(f"{'no S' if opt else 'with S'} {qps:7.0f} q/s p50 {p50:6.1f}")
        print(f"N={n} d={d} k={k}, {T:2d} callers: " + " | ".join(row) + f" | served again: {ix.debug_counter(25)}", flush=True)
    ix.close()
