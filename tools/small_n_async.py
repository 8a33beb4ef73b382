#!/usr/bin/env python
"""Scan-only launches on a small shard (asynchronous calls: every scan is followed by its own
finalize kernel, nothing rides along): kernel durations under rocprofv3 --kernel-trace --stats."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lean_explore_amd.index import FlatIPIndex  # noqa: E402
from tests import helpers as H  # noqa: E402

n = int(sys.argv[1])
ix = FlatIPIndex.from_array(H.gauss(1234, n, 384))
tq = torch.from_numpy(H.gauss(5678, 1, 384)).cuda()
o = (torch.empty((1, 50), device="cuda"), torch.empty((1, 50), dtype=torch.int64, device="cuda"))
for i in range(3000):
    ix.search_device(tq, 50, *o, asynchronous=True)
ix.check()
