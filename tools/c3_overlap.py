#!/usr/bin/env python
"""Experiment: do consecutive 1024-query batches overlap when they are queued on different
streams with their own scratch? (config 3: the MFMA pass fills every CU, the four small
kernels around it are latency-bound.) Uses H index handles over the same corpus, one stream
each, batches dealt round-robin:  python tools/c3_overlap.py [H ...]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lean_explore_amd.index import FlatIPIndex  # noqa: E402

N, D, NQ, K = 200_000, 384, 1024, 100
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
corpus = torch.randn(N, D, device=dev, generator=g, dtype=torch.float32)
corpus /= corpus.norm(dim=1, keepdim=True)
qs = [torch.randn(NQ, D, device=dev, generator=g, dtype=torch.float32) for _ in range(4)]
for q in qs:
    q /= q.norm(dim=1, keepdim=True)

for H in [int(a) for a in sys.argv[1:]] or [1, 2, 3]:
    idx = [FlatIPIndex.from_device_tensor(corpus, dtype="f16") for _ in range(H)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(H)]
    outs = [(torch.empty(NQ, K, device=dev), torch.empty(NQ, K, dtype=torch.int64, device=dev))
            for _ in range(H)]

    def run(steps):
        for i in range(steps):
            h = i % H
            idx[h].search_device(qs[i % 4], K, outs[h][0], outs[h][1], asynchronous=True,
                                 stream=streams[h])
        for h in range(H):
            idx[h].check(stream=streams[h])
        torch.cuda.synchronize()

    run(60)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        run(300)
        best = min(best, (time.perf_counter() - t0) / 300)
    ref_s, ref_i = idx[0].search_device(qs[0], K)
    torch.cuda.synchronize()
    print(f"handles/streams {H}: {best * 1e6:7.1f} us per batch  {NQ / best / 1e6:.3f} M queries/s", flush=True)
    del idx
