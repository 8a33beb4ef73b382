#!/usr/bin/env python
"""Phases of the config-3 SAMPLE pass's workgroups (variant build -DLS_GEMM_TIMING, 100 MHz stamps: start,
loads landed (query fragments + sample tiles), tiles computed, records written):
  LEANSEARCH_LIB=.../libleansearch_gtime.so python tools/sample_phases.py"""
import sys
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
for _ in range(10):
    ix.search_device(q, 100, asynchronous=True)
ix.check()
for rep in range(3):
    ix.search_device(q, 100, asynchronous=True); ix.check()
    v = np.array([ix.debug_counter(3000 + j) for j in range(4 * 256)], dtype=np.int64).reshape(-1, 4)
    t0 = v[:, 0].min()
    f = lambda a: " ".join(f"{x:.1f}" for x in np.percentile(a / 100.0, [0, 50, 100]))
    print(f"sample pass, 256 workgroups (us; min p50 max): start {f(v[:,0]-t0)} | start->loads landed {f(v[:,1]-v[:,0])} | "
          f"tiles {f(v[:,2]-v[:,1])} | records {f(v[:,3]-v[:,2])} | lifetime {f(v[:,3]-v[:,0])} | last end {(v[:,3].max()-t0)/100.0:.1f}", flush=True)
