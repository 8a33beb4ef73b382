#!/usr/bin/env python
"""Pipelined batch-1 scans with and without alternating the sweep direction of consecutive
launches (debug option 2): does the 256 MiB Infinity Cache serve the tail of the previous sweep?
  python tools/scan_alternate.py [n d dtype k]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lean_explore_amd.index import FlatIPIndex  # noqa: E402
from tests import helpers as H  # noqa: E402

a = sys.argv[1:]
n, d, dtype, k = (int(a[0]), int(a[1]), a[2], int(a[3])) if len(a) == 4 else (200_000, 384, "f32", 50)
c = H.gauss(1234, n, d)
q = H.gauss(5678, 1, d)
dev = torch.device("cuda:0")
tq = torch.from_numpy(q).to(dev)
ix = FlatIPIndex.from_array(c, dtype=dtype)
nbytes = n * d * (2 if dtype == "f16" else 4)
outs = [(torch.empty((1, k), dtype=torch.float32, device=dev),
         torch.empty((1, k), dtype=torch.int64, device=dev)) for _ in range(4)]
ref = ix.search(q, k)
K = 4000
for rnd in range(3):
    for alt in (0, 1):
        ix.debug_option(2, alt)
        for i in range(400):
            ix.search_device(tq, k, *outs[i % 4], pipeline=True)
        ix.check()
        t0 = time.perf_counter()
        for i in range(K):
            ix.search_device(tq, k, *outs[i % 4], pipeline=True)
        ix.check()
        dt = (time.perf_counter() - t0) / K
        ok = bool((outs[(K - 1) % 4][1].cpu().numpy() == ref[1]).all())
        print(f"round {rnd} alternate={alt}: {dt * 1e6:.2f} us/step  {nbytes / dt / 1e12:.3f} TB/s "
              f"({nbytes / dt / 8e12 * 100:.1f} % of 8 TB/s)  same result as plain search: {ok}", flush=True)
