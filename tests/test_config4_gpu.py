"""BASELINE config 4 at its real per-GPU size: 12.5 M rows x 768 fp16 (19.2 GB in HBM), 256
queries, k = 100, through the batched MFMA path. The oracle cannot hold such a shard, so the
checks are size-independent properties: a torch fp32 reference (fp16-rounded operands, blocked)
on 16 of the 256 queries, and shard-merge invariance (two half shards + ls_merge_topk == the full
shard, bit for bit)."""

import ctypes

import numpy as np
import pytest
import torch

from lean_explore_amd import native
from lean_explore_amd.index import FlatIPIndex

pytestmark = pytest.mark.gpu

ROWS, D, NQ, K = 12_500_000, 768, 256, 100
C4_RANK_DIFFS = 2  # measured on the round-6 library for seeds 1234 / 5678: 2 of 1600 ranks (all 1600 rows shared)


def _torch_reference(shard, tq, base, nv):
    q16 = tq[:nv].half().float()
    best_s = torch.full((nv, 0), 0.0, device=shard.device)
    best_i = torch.zeros((nv, 0), dtype=torch.int64, device=shard.device)
    for r0 in range(0, shard.shape[0], 1 << 20):
        sc = q16 @ shard[r0:r0 + (1 << 20)].half().float().T
        ts, ti = sc.topk(min(K, sc.shape[1]), dim=1)
        best_s = torch.cat([best_s, ts], 1)
        best_i = torch.cat([best_i, ti + r0 + base], 1)
        ts, sel = best_s.topk(min(K, best_s.shape[1]), dim=1)
        best_s, best_i = ts, best_i.gather(1, sel)
    return best_s.cpu().numpy(), best_i.cpu().numpy()


def test_config4_full_shard():
    free, _ = torch.cuda.mem_get_info()
    if free < 110 * (1 << 30):
        pytest.skip("needs ~100 GB of free HBM")
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    shard = torch.empty((ROWS, D), dtype=torch.float32, device=dev)
    for r0 in range(0, ROWS, 1 << 20):
        blk = torch.randn((min(1 << 20, ROWS - r0), D), device=dev, generator=gen)
        shard[r0:r0 + blk.shape[0]] = blk / blk.norm(dim=1, keepdim=True)
    gen.manual_seed(5678)
    tq = torch.randn((NQ, D), device=dev, generator=gen)
    tq /= tq.norm(dim=1, keepdim=True)
    base = 3 * ROWS  # this shard is rank 3 of 8: global row ids
    full = FlatIPIndex.from_device_tensor(shard, dtype="f16", base=base)
    s, i = full.search_device(tq, K, asynchronous=True)
    full.check()
    repaired = full.debug_counter(8)
    S, I = s.cpu().numpy(), i.cpu().numpy()
    # (1) torch reference on 16 queries: same rows; scores within 2e-5 (different fp32 summation)
    rs, ri = _torch_reference(shard, tq, base, 16)
    assert np.allclose(S[:16], rs, rtol=0, atol=2e-5)
    hits = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(I[:16], ri))
    assert hits >= 16 * K - 2, hits  # a near-tie at the k-th place may legitimately differ
    # (round 6: what holds today for these seeds is asserted - the same ROW SETS, and how many of the 1600 ranks
    # hold another row than the torch reference's, near ties swapped by the summation order)
    rank_diffs = int((I[:16] != ri).sum())
    print("config4 full shard: rows shared", hits, "of", 16 * K, "; ranks that differ", rank_diffs, "; repaired", repaired)
    assert hits == 16 * K and rank_diffs <= C4_RANK_DIFFS, (hits, rank_diffs)
    # (2) structure: sorted by (score desc, row asc), rows inside the shard's global range
    assert np.all(S[:, :-1] >= S[:, 1:]) and I.min() >= base and I.max() < base + ROWS
    assert repaired <= 2, repaired  # exchangeable rows: a repair is a ~1e-5 event per query
    # (3) shard-merge invariance: two half shards merged by ls_merge_topk == the full shard
    half = ROWS // 2
    lo = FlatIPIndex.from_device_tensor(shard[:half], dtype="f16", base=base)
    hi = FlatIPIndex.from_device_tensor(shard[half:].contiguous(), dtype="f16", base=base + half)
    del shard
    torch.cuda.empty_cache()
    ps = torch.empty((2, NQ, K), dtype=torch.float32, device=dev)
    pi = torch.empty((2, NQ, K), dtype=torch.int64, device=dev)
    lo.search_device(tq, K, ps[0], pi[0], asynchronous=True)
    hi.search_device(tq, K, ps[1], pi[1], asynchronous=True)
    lo.check()
    hi.check()
    ms = torch.empty((NQ, K), dtype=torch.float32, device=dev)
    mi = torch.empty((NQ, K), dtype=torch.int64, device=dev)
    native.check(native.load().ls_merge_topk(ps.data_ptr(), pi.data_ptr(), 2, NQ, K, ms.data_ptr(),
                                             mi.data_ptr(), 0,
                                             torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert np.array_equal(ms.cpu().numpy(), S) and np.array_equal(mi.cpu().numpy(), I)
    for ix in (full, lo, hi):
        ix.close()
