#!/usr/bin/env python
"""Phases of config 3's FUSED filter launch (variant build -DLS_GEMM_TIMING, 100 MHz stamps per workgroup):
  make -C lean-explore_amd/csrc variant NAME=gtime VFLAGS=-DLS_GEMM_TIMING
  LEANSEARCH_LIB=lean-explore_amd/variants/libleansearch_gtime.so python tools/fused_phases.py
Also prints the host time one pipelined call takes to queue."""
import sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lean_explore_amd.index import FlatIPIndex  # noqa: E402
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
c = torch.randn(200_000, 384, device=dev, generator=g); c /= c.norm(dim=1, keepdim=True)
q = torch.randn(1024, 384, device=dev, generator=g); q /= q.norm(dim=1, keepdim=True)
ix = FlatIPIndex.from_device_tensor(c, dtype="f16")
outs = [(torch.empty((1024, 100), device=dev), torch.empty((1024, 100), dtype=torch.int64, device=dev)) for _ in range(16)]
for mode in ("asynchronous", "pipeline"):
    for _ in range(10):
        ix.search_device(q, 100, *outs[0], **{mode: True})
    ix.check()
    t = []
    for i in range(64):
        t0 = time.perf_counter()
        ix.search_device(q, 100, *outs[i % 16], **{mode: True})
        t.append(time.perf_counter() - t0)
        if i % 16 == 15:
            ix.check()
    print(f"{mode}: host time to queue one call: p50 {np.median(t)*1e6:.1f} us, p90 {np.quantile(t, .9)*1e6:.1f} us", flush=True)
for rep in range(3):
    for _ in range(4):
        ix.search_device(q, 100, *outs[0], pipeline=True)
    ix.check()
    v = np.array([ix.debug_counter(3000 + j) for j in range(8 * 256)], dtype=np.int64).reshape(-1, 8)
    t0 = v[:, 0].min()
    f = lambda a: "/".join(f"{x:.1f}" for x in np.percentile(a / 100.0, [0, 50, 100]))
    print("pass + riding sample phase, 256 workgroups of the last fused launch (us; min/p50/max): start " + f(v[:, 0] - t0) +
          f" | full pass {f(v[:, 2] - v[:, 0])} | sample phase {f(v[:, 7] - v[:, 2])} | last end {(v[:, 7].max() - t0) / 100.0:.1f}", flush=True)
