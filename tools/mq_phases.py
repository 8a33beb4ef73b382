#!/usr/bin/env python
"""Phase times of one ls_mq workgroup (variant build with -DLS_SCAN_TIMING; 100 MHz ticks):
   make -C lean-explore_amd/csrc variant NAME=stime VFLAGS=-DLS_SCAN_TIMING
   LEANSEARCH_LIB=.../variants/libleansearch_stime.so python tools/mq_phases.py"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lean_explore_amd.index import FlatIPIndex  # noqa: E402
from tests import helpers as H  # noqa: E402

for n, d, k in ((200_000, 384, 50), (200_000, 1024, 50), (25_000, 384, 50)):
    ix = FlatIPIndex.from_array(H.gauss(1234, n, d))
    for nq in (2, 16):
        acc = np.zeros(7)
        for i in range(30):
            ix.search(H.gauss(100 + i, nq, d), k)
            if i >= 10:
                acc += np.array([ix.debug_counter(10 + j) for j in range(7)])
        acc = acc / 20
        us = acc[:6] / 100.0
        print(f"N={n} d={d} nq={nq}: query staging {us[0]:.2f} us | first task: units+tree+park {us[1]:.2f}, epilogue {us[2]:.2f} | "
              f"remaining {acc[6] - 1:.0f} tasks {us[3]:.2f} | lists->LDS+barrier {us[4]:.2f} | rank+emit {us[5]:.2f}", flush=True)
    ix.close()
