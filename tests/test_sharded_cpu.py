"""The N>1 path on CPU: row partition, the all-gather exchange (gloo, world_size 2 and 3) and the
merge order. The HIP kernels cannot run here, so the two compute hooks of ShardedFlatIPIndex are
replaced by the CPU oracle (tests may call the oracle; the product defaults are HIP-only)."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lean_explore_amd.sharded import ShardedFlatIPIndex, shard_bounds
from oracle import oracle
from tests import helpers as H


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 7, 8, 9, 200_000, 100_000_001):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans[:-1], spans[1:]):
                assert b == c and a <= b
            per = -(-n // w) if n else 0
            assert all(b - a <= per for a, b in spans)
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


class _FakeLocal:
    """Stands in for FlatIPIndex on a CPU-only host: same attributes the sharded layer uses."""

    def __init__(self, rows, base):
        self.rows, self.base, self.d, self.device = rows, base, rows.shape[1], 0


def _worker(rank, world, port, n, d, nq, k, seed, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        corpus = H.int_corpus(seed, n, d)      # exact arithmetic: results must be bit-identical
        q = H.int_corpus(seed + 1, nq, d)
        lo, hi = shard_bounds(n, world, rank)
        local = _FakeLocal(corpus[lo:hi], lo)

        def local_search(tq, kk, normalize):
            D, I = oracle.c_search(local.rows, tq.numpy(), kk, base=local.base, normalize=normalize)
            return torch.from_numpy(D), torch.from_numpy(I)

        def merge(all_s, all_i, kk):
            D, I = oracle.c_merge(all_s.numpy(), all_i.numpy())
            return torch.from_numpy(D), torch.from_numpy(I)

        ix = ShardedFlatIPIndex(local, n, local_search=local_search, merge=merge)
        assert ix.world == world and ix.rank == rank and ix.ntotal == n and ix.d == d
        s, i = ix.search_device(torch.from_numpy(q), k)
        Dref, Iref = oracle.c_search(corpus, q, k)
        ok = np.array_equal(s.numpy(), Dref) and np.array_equal(i.numpy(), Iref)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n,k", [(2, 1001, 40), (3, 500, 64), (2, 30, 50)])
def test_gloo_sharded_search_equals_unsharded(world, n, k):
    """Every rank ends with the 1-GPU answer, including k > rows-per-shard (-1 padded shards)."""
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, n, 32, 3, k, 77, ret))
                 for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert dict(ret) == {r: True for r in range(world)}
