"""The sharded path end to end with REAL HIP kernels and world_size 2 / 3 on ONE GPU: every rank
runs the HIP local search on its row block, the packed [scores | rows] exchange goes through
torch.distributed (gloo here: RCCL cannot place two ranks on one device; the collective call and
buffer layout are the ones the nccl backend receives), and the HIP strided merge produces the
final answer, which must equal the unsharded oracle on every rank."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, cfg, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lean_explore_amd.sharded import ShardedFlatIPIndex
        from oracle import oracle

        n, d, nq, k, dtype, ints = cfg
        torch.cuda.set_device(0)
        if ints:
            corpus, q = H.int_corpus(5, n, d), H.int_corpus(6, nq, d)
        else:
            corpus, q = H.gauss(5, n, d), H.gauss(6, nq, d)
        ix = ShardedFlatIPIndex.from_array(corpus, dtype=dtype, device=0)
        assert ix.world == world and ix.local.base == (rank * (-(-n // world)))
        D, I = ix.search(q, k)
        Dr, Ir = oracle.c_search(corpus, q, k, f16=(dtype == "f16"))
        if ints:
            ok = np.array_equal(D, Dr) and np.array_equal(I, Ir)
        else:
            _, _, S = oracle.np_search(corpus, q, k, f16=(dtype == "f16"))
            ok = oracle.compare_topk(D, I, Dr, Ir, S)["recall"] == 1.0
        # the pipelined mode: a stream of single queries, grouped M per exchange, results valid
        # after flush(); every step still in the ring (depth 4 groups) is checked
        if nq <= 16:
            tq = torch.from_numpy(q).cuda()
            for M, steps in ((1, 7), (3, 11), (8, 21)):
                outs = []
                for j in range(steps):
                    qq = tq[j % nq: j % nq + 1]
                    s_, i_ = ix.search_device_pipelined(qq, k, exchange_every=M)
                    outs.append((j % nq, j // M, s_, i_))
                ix.flush()
                newest = (steps - 1) // M
                for (qi, grp, s_, i_) in outs:
                    if grp < newest - 3:
                        continue  # overwritten in the ring
                    got_s, got_i = s_.cpu().numpy(), i_.cpu().numpy()
                    if ints:
                        ok = ok and np.array_equal(got_s, Dr[qi:qi + 1]) \
                            and np.array_equal(got_i, Ir[qi:qi + 1])
                    else:
                        ok = ok and oracle.compare_topk(got_s, got_i, Dr[qi:qi + 1], Ir[qi:qi + 1],
                                                        S[qi:qi + 1])["recall"] == 1.0
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,cfg", [
    (2, (60_000, 384, 3, 50, "f32", False)),      # config-2 shape, sharded 2 ways
    (3, (50_001, 128, 5, 100, "f32", True)),      # ragged shards, exact integer data, ties
    (2, (90_000, 384, 64, 100, "f16", False)),    # batched MFMA path per shard (sync + repair)
    (2, (300, 64, 2, 1000, "f32", True)),         # k > rows per shard: -1 padded shard lists
    (3, (7_777, 200, 4, 300, "f32", True)),       # odd sizes, k > 256 (bitonic finalize), ties
    (2, (70_000, 768, 160, 100, "f16", True)),    # long rows, batched path per shard, exact ints
    (3, (30_000, 384, 130, 64, "f16", False)),    # small shards (10k rows): batched via BIGNQ rule
])
def test_two_ranks_one_gpu(world, cfg):
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, cfg, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        assert dict(ret) == {r: True for r in range(world)}


def _nccl_single(port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from lean_explore_amd.sharded import ShardedFlatIPIndex
        from oracle import oracle

        corpus, q = H.int_corpus(9, 40_000, 256), H.int_corpus(10, 4, 256)
        ix = ShardedFlatIPIndex.from_array(corpus, device=0)
        ix.force_exchange = True  # the packed all-gather + strided merge really run, over RCCL
        D, I = ix.search(q, 77)
        Dr, Ir = oracle.c_search(corpus, q, 77)
        ok = np.array_equal(D, Dr) and np.array_equal(I, Ir)
        tq = torch.from_numpy(q).cuda()
        outs = [ix.search_device_pipelined(tq[j % 4: j % 4 + 1], 77, exchange_every=4)
                for j in range(10)]
        ix.flush()
        for j in range(10):
            ok = ok and np.array_equal(outs[j][0].cpu().numpy(), Dr[j % 4: j % 4 + 1]) \
                and np.array_equal(outs[j][1].cpu().numpy(), Ir[j % 4: j % 4 + 1])
        ret[0] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_rccl_exchange_single_rank():
    """backend "nccl" (= RCCL): the packed uint8 all-gather and the merge on one rank."""
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        p = ctx.Process(target=_nccl_single, args=(_free_port(), ret))
        p.start()
        p.join(300)
        assert p.exitcode == 0 and dict(ret) == {0: True}
