#!/usr/bin/env python
"""Start / end time of every scan workgroup of one launch (variant build with -DLS_SCAN_TIMING):
is the HBM stream saturated until the end, or do workgroups with a static share of the tiles finish
far apart?  LEANSEARCH_LIB=.../libleansearch_stime.so python tools/scan_lifetimes.py [n d]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lean_explore_amd.index import FlatIPIndex  # noqa: E402
from tests import helpers as H  # noqa: E402

n, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200_000, 384)
ix = FlatIPIndex.from_array(H.gauss(1234, n, d))
for i in range(20):
    ix.search(H.gauss(100 + i, 1, d), 50)
blocks = 448 if n >= 200_000 else 256
for rep in range(3):
    ix.search(H.gauss(500 + rep, 1, d), 50)
    v = np.array([ix.debug_counter(1000 + j) for j in range(2 * 512)], dtype=np.int64).reshape(-1, 2)
    live = v[(v[:, 0] > 0) & (v[:, 1] > v[:, 0])]
    t0 = live[:, 0].min()
    st, en = (live[:, 0] - t0) / 100.0, (live[:, 1] - t0) / 100.0
    q = lambda a: " ".join(f"{x:.1f}" for x in np.percentile(a, [0, 10, 50, 90, 100]))
    print(f"N={n} d={d}: {len(live)} workgroups; start us (min p10 p50 p90 max): {q(st)}; "
          f"end us: {q(en)}; lifetime: {q(en - st)}", flush=True)
    by_xcd = [en[np.arange(len(en)) % 8 == x].mean() for x in range(8)]
    print("   mean end per blockIdx % 8:", " ".join(f"{x:.1f}" for x in by_xcd), flush=True)
