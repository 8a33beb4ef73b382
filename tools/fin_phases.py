#!/usr/bin/env python
"""Phases of the selection step (finalize_body) of a synchronous host search, variant build -DLS_FIN_TIMING:
  make -C lean-explore_amd/csrc variant NAME=ftime VFLAGS=-DLS_FIN_TIMING
  LEANSEARCH_LIB=lean-explore_amd/variants/libleansearch_ftime.so python tools/fin_phases.py
100 MHz stamps of the selection workgroup: pivot found | pre-filtered keys in LDS | k-th key found (radix passes; lists of
257..4096 keys: sample ranked + bucket counts + scan of the splitter buckets) | survivors compacted (scattered to their
buckets) | ordered | outputs written + completion word."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lean_explore_amd.index import FlatIPIndex  # noqa: E402
from tests import helpers as H  # noqa: E402

for (n, d, k) in [(200_000, 384, 50), (200_000, 1024, 1000)]:
    c = H.gauss(1234, n, d); q = H.gauss(5678, 1, d)
    ix = FlatIPIndex.from_array(c)
    for mode in (1, 0):
        ix.debug_option(9, mode)
        for _ in range(20):
            ix.search(q, k, normalize=True)
        ph = []
        lat = []
        for _ in range(50):
            t0 = time.perf_counter(); ix.search(q, k, normalize=True); lat.append(time.perf_counter() - t0)
            ph.append([ix.debug_counter(2 + i) for i in range(6)] + [ix.debug_counter(1)])
        ph = np.median(np.array(ph), axis=0) / 100.0
        print(f"N={n} d={d} k={k} selection {'inside the scan launch' if mode else 'as its own launch'}: call p50 {np.median(lat)*1e6:.1f} us | "
              f"loads issued + pivot {ph[0]:.1f} (pivot loads landed after {ph[6]:.1f}) | survivors to LDS {ph[1]:.1f} | k-th key {ph[2]:.1f} | compact {ph[3]:.1f} | "
              f"order {ph[4]:.1f} | output+done {ph[5]:.1f} us", flush=True)
    ix.close()
