#!/usr/bin/env python
"""Timeline of one synchronous host search whose selection rides inside the scan launch (variant build):
  make -C lean-explore_amd/csrc variant NAME=hot VFLAGS=-DLS_HANDOFF_TIMING
  LEANSEARCH_LIB=lean-explore_amd/variants/libleansearch_hot.so python tools/handoff_timeline.py
All device times are us after the selection workgroup's own entry into the kernel (100 MHz clock): first scan
workgroup started | last scan workgroup ended (its granule stores issued) | every granule swept | keys extracted,
pivot plane in the pivot wave's registers | pivot found | survivors in LDS | outputs + completion word written. The host's call latency minus the
last figure is launch latency + the completion word's way to the polling core."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lean_explore_amd.index import FlatIPIndex  # noqa: E402
from tests import helpers as H  # noqa: E402

for (n, d, k, dt) in [(200_000, 384, 50, "f32"), (25_000, 384, 50, "f32"), (200_000, 384, 50, "f16")]:
    c = H.gauss(1234, n, d); q = H.gauss(5678, 1, d)
    ix = FlatIPIndex.from_array(c, dtype=dt)
    ix.debug_option(9, 1)
    for _ in range(20):
        ix.search(q, k, normalize=True)
    rows, lat = [], []
    for _ in range(60):
        t0 = time.perf_counter(); ix.search(q, k, normalize=True); lat.append(time.perf_counter() - t0)
        rows.append([ix.debug_counter(1 + i) for i in range(7)])
    r = (np.median(np.array(rows, dtype=np.float64), axis=0) - 10000.0) / 100.0
    print(f"N={n} d={d} {dt} k={k}: call p50 {np.median(lat)*1e6:.1f} us | first scan wg start {r[0]:+.1f} | last scan wg end {r[1]:.1f} | "
          f"swept {r[2]:.1f} | pivot plane in registers {r[3]:.1f} | pivot found {r[4]:.1f} | survivors {r[5]:.1f} | done {r[6]:.1f}", flush=True)
    ix.close()
