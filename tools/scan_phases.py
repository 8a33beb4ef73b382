#!/usr/bin/env python
"""Phase times of one scan workgroup on small shards (variant build with -DLS_SCAN_TIMING; 100 MHz
ticks): LEANSEARCH_LIB=.../libleansearch_stime.so python tools/scan_phases.py"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lean_explore_amd.index import FlatIPIndex  # noqa: E402
from tests import helpers as H  # noqa: E402

for n in (10_000, 25_000, 50_000, 200_000):
    ix = FlatIPIndex.from_array(H.gauss(1234, n, 384))
    acc = np.zeros(4)
    for i in range(40):
        ix.search(H.gauss(100 + i, 1, 384), 50)
        if i >= 10:
            acc += np.array([ix.debug_counter(10 + j) for j in range(4)])
    acc = acc / 30 / 100.0  # us
    print(f"N={n}: query prep {acc[0]:.2f} us, tile loop {acc[1]:.2f}, wave lists -> LDS + barrier {acc[2]:.2f}, "
          f"merge + stores {acc[3]:.2f}  (one workgroup in the middle of the grid)", flush=True)
    ix.close()
