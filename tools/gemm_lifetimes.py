#!/usr/bin/env python
"""Start / end time of every workgroup of one MFMA pass (variant build with -DLS_GEMM_TIMING).
  LEANSEARCH_LIB=.../libleansearch_gtime.so python tools/gemm_lifetimes.py c3|c4"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lean_explore_amd.index import FlatIPIndex  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
n, d, nq, k = (200_000, 384, 1024, 100) if wl == "c3" else (4_000_000, 768, 256, 100)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
c = torch.randn(n, d, device=dev, generator=g)
c /= c.norm(dim=1, keepdim=True)
q = torch.randn(nq, d, device=dev, generator=g)
q /= q.norm(dim=1, keepdim=True)
ix = FlatIPIndex.from_device_tensor(c, dtype="f16")
del c
for _ in range(20):
    ix.search_device(q, k, asynchronous=True)
ix.check()
for rep in range(3):
    ix.search_device(q, k, asynchronous=True)
    ix.check()
    v = np.array([ix.debug_counter(2000 + j) for j in range(2 * 256)], dtype=np.int64).reshape(-1, 2)
    t0 = v[:, 0].min()
    st, en = (v[:, 0] - t0) / 100.0, (v[:, 1] - t0) / 100.0
    qs = lambda a: " ".join(f"{x:.1f}" for x in np.percentile(a, [0, 10, 50, 90, 100]))
    print(f"{wl}: 256 workgroups; start us (min p10 p50 p90 max): {qs(st)}; end us: {qs(en)}; lifetime: {qs(en - st)}")
    print("   mean lifetime per blockIdx % 8 (XCD):", " ".join(f"{(en - st)[np.arange(256) % 8 == x].mean():.1f}" for x in range(8)), flush=True)
