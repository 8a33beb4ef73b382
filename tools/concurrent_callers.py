"""Throughput of T threads making synchronous single-query ls_search calls on ONE handle (ctypes
releases the GIL), with and without combining (debug option 10):
    gpurun -- 'python tools/concurrent_callers.py'"""
import sys, threading, time; sys.path.insert(0, '/root/repo')
import numpy as np
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H

from tools.concurrent_callers_lib import run

for n in (200_000, 25_000):
    c = H.gauss(1234, n, 384); q = H.gauss(5678, 16, 384)
    ix = FlatIPIndex.from_array(c)
    for _ in range(50): ix.search(q[:1], 50, normalize=True)
    for T in (1, 2, 4, 8, 16):
        row = []
        for comb, ov in ((1, 1), (1, 0), (0, 1)):
            ix.debug_option(10, comb)
            ix.debug_option(17, ov)  # synchronous calls may overlap two deep (round 5)
            qps, p50 = run(ix, q, 50, T)
            row.append(f"{'combined' if comb else 'serialised'}{'' if ov else ' (no overlap)'} {qps:8.0f} q/s p50 {p50:6.1f} us")
        ix.debug_option(17, 1)
        print(f"N={n} d=384 f32 k=50, {T:2d} callers: " + " | ".join(row), flush=True)
    print("   combined batches", ix.debug_counter(16), "requests in them", ix.debug_counter(17), flush=True)
    ix.close()

# Replicas (ls_create_replicated): every visible GPU holds the whole corpus, synchronous calls are dealt
# round-robin. On a one-GPU box the replicas share the device (a rehearsal of the mechanism: the scans then
# share one HBM, so there is nothing to gain); on a multi-GPU node this is the number that scales.
import torch
ndev = max(1, torch.cuda.device_count())
devs = list(range(ndev)) if ndev > 1 else [0, 0]
c = H.gauss(1234, 200_000, 384); q = H.gauss(5678, 16, 384)
ix = FlatIPIndex.from_array(c, devices=devs, replicate=True)
for _ in range(50): ix.search(q[:1], 50, normalize=True)
for T in (1, 2, 4, 8, 16):
    qps, p50 = run(ix, q, 50, T)
    print(f"N=200000 d=384 f32 k=50, {len(devs)} replicas on devices {devs}, {T:2d} callers: {qps:8.0f} q/s p50 {p50:6.1f} us", flush=True)
ix.close()
