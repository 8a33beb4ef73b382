#!/usr/bin/env python
"""Start / end time of every ls_mq workgroup of one launch (variant build with -DLS_SCAN_TIMING): how many
workgroups run at once?  LEANSEARCH_LIB=.../libleansearch_stime.so python tools/mq_lifetimes.py [n d]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lean_explore_amd.index import FlatIPIndex  # noqa: E402
from tests import helpers as H  # noqa: E402

n, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200_000, 384)
ix = FlatIPIndex.from_array(H.gauss(1234, n, d))
ix.debug_option(19, 0)  # keep the score vectors: the lifetimes are parked in score vector 7 (S must exist)
for i in range(20):
    ix.search(H.gauss(100 + i, 4, d), 50)
for rep in range(2):
    ix.search(H.gauss(500 + rep, 4, d), 50)
    v = np.array([ix.debug_counter(1000 + j) for j in range(2 * 512)], dtype=np.int64).reshape(-1, 2)
    live = v[(v[:, 0] > 0) & (v[:, 1] > v[:, 0])]
    t0 = live[:, 0].min()
    st, en = (live[:, 0] - t0) / 100.0, (live[:, 1] - t0) / 100.0
    q = lambda a: " ".join(f"{x:.1f}" for x in np.percentile(a, [0, 10, 25, 50, 75, 90, 100]))
    print(f"N={n} d={d}: {len(live)} workgroups; start us (min p10 p25 p50 p75 p90 max): {q(st)}; "
          f"end us: {q(en)}; lifetime: {q(en - st)}", flush=True)
    for tt in (2, 5, 10, 20, 30, 40, 50, 60):
        print(f"   running at t={tt} us: {int(((st <= tt) & (en > tt)).sum())}", end=";")
    print(flush=True)
