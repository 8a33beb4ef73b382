"""The in-library row-sharded handle (ls_create_sharded): ONE process, one ls_index* over several
devices, every entry point of leansearch.h unchanged (reference process model
mcp/server.py:147-151, call site search/engine.py:250).

A one-GPU box rehearses G shards with device_ids = [0] * G (the exchange is then a device-to-device
copy: RCCL cannot put two ranks on one device); the RCCL exchange itself is exercised with one
rank (device_ids = [0]) and, whenever the box has >= 2 GPUs, with every visible device.
Everything is compared with the UNSHARDED CPU oracle: bit-exact on integer corpora (ties
included), compare_topk's near-tie rule on continuous data."""

import ctypes

import numpy as np
import pytest

from lean_explore_amd import native
from lean_explore_amd.index import FlatIPIndex
from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _check(D, I, corpus, q, k, f16, ints, normalize=False):
    Dr, Ir = oracle.c_search(corpus, q, k, normalize=normalize, f16=f16)
    if ints and not normalize:
        np.testing.assert_array_equal(I, Ir)
        np.testing.assert_array_equal(D, Dr)
        return
    qn = oracle.c_normalize_l2(q) if normalize else q
    _, _, S = oracle.np_search(corpus, qn, k, f16=f16)
    rep = oracle.compare_topk(D, I, Dr, Ir, S)
    assert rep["recall"] == 1.0 and rep["max_score_err"] <= 1e-5, rep


@pytest.mark.parametrize("G,n,d,nq,k,dtype,ints", [
    (3, 50_001, 128, 5, 100, "f32", True),      # ragged blocks, exact integer data, thousands of ties
    (3, 60_000, 384, 1, 50, "f32", False),      # config-2 shape, the reference's nq = 1
    (2, 40_000, 384, 9, 1000, "f32", False),    # the reference's k = 1000
    (4, 30_000, 64, 3, 50, "f16", True),        # fp16 storage
    (3, 2_000, 96, 4, 1500, "f32", True),       # k > rows per shard: every shard pads with -1
    (5, 3, 32, 2, 10, "f32", True),             # n < G: empty shards, k > n
    (8, 20_000, 64, 2, 2048, "f32", True),      # G * k beyond one merge launch: merge in rounds
])
def test_sharded_handle_matches_unsharded_oracle(G, n, d, nq, k, dtype, ints):
    corpus = H.int_corpus(5, n, d) if ints else H.gauss(5, n, d)
    q = H.int_corpus(6, nq, d) if ints else H.gauss(6, nq, d)
    ix = FlatIPIndex.from_array(corpus, dtype=dtype, devices=[0] * G)
    try:
        sh = ix.shards()
        per = -(-n // G)
        assert len(sh) == G and sum(r for _, _, r in sh) == n
        assert [s[1] for s in sh] == [min(n, g * per) for g in range(G)]
        assert ix.ntotal == n and ix.d == d
        D, I = ix.search(q, k)                       # the host API (ls_search) on the group handle
        _check(D, I, corpus, q, k, dtype == "f16", ints)
        assert ix.debug_counter(13) >= 1             # an exchange step really ran
        assert ix.debug_counter(15) == 0             # duplicate ids: copies, not RCCL
    finally:
        ix.close()


def test_sharded_handle_device_api_async_and_normalize():
    import torch

    n, d, nq, k = 45_000, 384, 7, 64
    corpus, q = H.gauss(11, n, d), H.gauss(12, nq, d, normalize=False)
    ix = FlatIPIndex.from_array(corpus, devices=[0, 0, 0])
    try:
        tq = torch.from_numpy(q).cuda()
        side = torch.cuda.Stream()
        outs = []
        for j in range(6):  # consecutive async calls, alternating streams, reusing the scan slot
            st = side if j & 1 else torch.cuda.current_stream()
            with torch.cuda.stream(st):
                s, i = ix.search_device(tq, k, normalize=True, asynchronous=True, stream=st)
            outs.append((s, i))
        ix.check()
        torch.cuda.synchronize()
        for s, i in outs:
            _check(s.cpu().numpy(), i.cpu().numpy(), corpus, q, k, False, False, normalize=True)
        # synchronous device call and the pipelined flag (scan path: treated as async)
        s, i = ix.search_device(tq, k, normalize=True)
        _check(s.cpu().numpy(), i.cpu().numpy(), corpus, q, k, False, False, normalize=True)
        s, i = ix.search_device(tq, k, normalize=True, pipeline=True)
        ix.check()
        _check(s.cpu().numpy(), i.cpu().numpy(), corpus, q, k, False, False, normalize=True)
    finally:
        ix.close()


@pytest.mark.parametrize("dtype,nq,k", [("f16", 64, 100), ("f16", 300, 100), ("f32", 64, 1000)])
def test_sharded_handle_batched_paths(dtype, nq, k):
    """Every shard answers through its speculative MFMA path (>= 32 768 rows per shard); flags
    travel with the packed blocks; synchronous, async + check and pipelined calls."""
    import torch

    n, d = 100_000, 256
    corpus, q = H.gauss(21, n, d), H.gauss(22, nq, d)
    ix = FlatIPIndex.from_array(corpus, dtype=dtype, devices=[0, 0, 0])
    try:
        D, I = ix.search(q, k)
        assert ix.debug_counter(10) in (2, 3)        # the primary shard took an MFMA path
        _check(D, I, corpus, q, k, dtype == "f16", False)
        tq = torch.from_numpy(q).cuda()
        outs = [ix.search_device(tq, k, asynchronous=True) for _ in range(3)]
        outs += [ix.search_device(tq, k, pipeline=True) for _ in range(11)]  # > LS_SH_SLOTS
        ix.check()
        for s, i in outs:
            _check(s.cpu().numpy(), i.cpu().numpy(), corpus, q, k, dtype == "f16", False)
    finally:
        ix.close()


def test_sharded_handle_repair_after_the_exchange():
    """A planted cluster in ONE shard overflows that shard's candidate queues: its flags travel
    with the results, ls_check repairs the shard's rows in place and the group exchanges and
    merges again into the same output tensors."""
    import torch

    n, d, nq, k = 120_000, 128, 64, 100
    rng = np.random.default_rng(3)
    corpus = H.gauss(31, n, d)
    q = H.gauss(32, nq, d)
    # 6000 near-copies of query 5 inside shard 1's block (rows 40 000 ..): far more rows pass the
    # sample threshold than one slice's queue holds
    rows = 40_000 + rng.choice(40_000, size=6000, replace=False)
    corpus[rows] = q[5] + 0.01 * rng.standard_normal((6000, d)).astype(np.float32)
    corpus[rows] /= np.linalg.norm(corpus[rows], axis=1, keepdims=True)
    ix = FlatIPIndex.from_array(corpus, dtype="f16", devices=[0, 0, 0])
    try:
        tq = torch.from_numpy(q).cuda()
        s, i = ix.search_device(tq, k, asynchronous=True)
        ix.check()
        assert ix.debug_counter(8) >= 1              # some shard repaired a query ...
        assert ix.debug_counter(14) >= 1             # ... and the group exchanged again
        _check(s.cpu().numpy(), i.cpu().numpy(), corpus, q, k, True, False)
        D, I = ix.search(q, k)                       # synchronous host call: repaired inside
        _check(D, I, corpus, q, k, True, False)
    finally:
        ix.close()


def test_sharded_handle_repair_done_early_by_a_shard_is_still_re_merged():
    """Call A (64 queries, one of them flagged in shard 1) stays unchecked; call B brings a BIGGER
    batch, which makes the shard re-slice its flag slots and repair A on the spot - after A's
    provisional rows were already exchanged and merged. The group's next check must notice (it
    compares the shards' repair counters with their values at the previous check) and merge A again."""
    import torch

    n, d, k = 120_000, 128, 100
    rng = np.random.default_rng(4)
    corpus = H.gauss(33, n, d)
    q = H.gauss(34, 320, d)
    rows = 40_000 + rng.choice(40_000, size=6000, replace=False)
    corpus[rows] = q[5] + 0.01 * rng.standard_normal((6000, d)).astype(np.float32)
    corpus[rows] /= np.linalg.norm(corpus[rows], axis=1, keepdims=True)
    ix = FlatIPIndex.from_array(corpus, dtype="f16", devices=[0, 0, 0])
    try:
        tq = torch.from_numpy(q).cuda()
        sA, iA = ix.search_device(tq[:64].contiguous(), k, asynchronous=True)
        sB, iB = ix.search_device(tq, k, asynchronous=True)
        ix.check()
        assert ix.debug_counter(8) >= 1
        _check(sA.cpu().numpy(), iA.cpu().numpy(), corpus, q[:64], k, True, False)
        _check(sB.cpu().numpy(), iB.cpu().numpy(), corpus, q, k, True, False)
    finally:
        ix.close()


def test_sharded_handle_add_reconstruct_base():
    n, d, k = 9_000, 48, 20
    corpus = H.int_corpus(41, n, d)
    more = H.int_corpus(42, 1_111, d)
    q = H.int_corpus(43, 3, d)
    ix = FlatIPIndex.from_array(corpus, devices=[0, 0, 0], base=1000)
    try:
        D, I = ix.search(q, k)
        Dr, Ir = oracle.c_search(corpus, q, k)
        np.testing.assert_array_equal(I, Ir + 1000)
        np.testing.assert_array_equal(D, Dr)
        ix.add(more)                                 # extends the last shard
        full = np.concatenate([corpus, more])
        assert ix.ntotal == n + 1_111 and ix.shards()[-1][2] == 3_000 + 1_111
        D, I = ix.search(q, k)
        Dr, Ir = oracle.c_search(full, q, k)
        np.testing.assert_array_equal(I, Ir + 1000)
        np.testing.assert_array_equal(D, Dr)
        np.testing.assert_array_equal(ix.host_corpus(), full)   # ls_reconstruct across shards
    finally:
        ix.close()


def test_sharded_handle_from_device_blocks():
    import torch

    d, k = 64, 30
    blocks = [H.int_corpus(50 + g, r, d) for g, r in enumerate((5_000, 1, 7_777))]
    corpus = np.concatenate(blocks)
    q = H.int_corpus(59, 4, d)
    ix = FlatIPIndex.from_device_blocks([torch.from_numpy(b).cuda() for b in blocks], dtype="f16")
    try:
        assert [(s[1], s[2]) for s in ix.shards()] == [(0, 5_000), (5_000, 1), (5_001, 7_777)]
        D, I = ix.search(q, k)
        _check(D, I, corpus, q, k, True, True)
    finally:
        ix.close()


def test_rccl_exchange_with_one_rank():
    """device_ids = [0] is a group of one: distinct ids, so the exchange step is the library's own
    ncclAllGather (communicator from ncclCommInitAll) — RCCL is bound, initialised and launched
    on a one-GPU box."""
    n, d, nq, k = 30_000, 128, 6, 77
    corpus, q = H.int_corpus(61, n, d), H.int_corpus(62, nq, d)
    ix = FlatIPIndex.from_array(corpus, devices=[0])
    try:
        assert len(ix.shards()) == 1
        D, I = ix.search(q, k)
        assert ix.debug_counter(15) == 2             # RCCL communicators initialised and used
        _check(D, I, corpus, q, k, False, True)
        assert "rccl all-gather" in ix.exchange_info()["exchange"]
        ix.debug_option(8, 2)                        # RCCL gather-to-root: ncclSend / ncclRecv, only the primary receives
        D, I = ix.search(q, k)
        assert ix.debug_counter(15) == 2
        assert "gather-to-root" in ix.exchange_info()["exchange"]
        _check(D, I, corpus, q, k, False, True)
        ix.debug_option(8, 1)                        # the same group over peer copies
        D, I = ix.search(q, k)
        assert ix.debug_counter(15) == 0
        _check(D, I, corpus, q, k, False, True)
    finally:
        ix.close()


def test_rccl_failure_falls_back_to_peer_copies():
    """If RCCL cannot be used on the node (here: a failure injected into the exchange of a one-rank
    group) the handle must keep answering through the copy exchange, and say so."""
    n, d, nq, k = 30_000, 128, 6, 77
    corpus, q = H.int_corpus(63, n, d), H.int_corpus(64, nq, d)
    ix = FlatIPIndex.from_array(corpus, devices=[0])
    try:
        ix.debug_option(12, 1)
        D, I = ix.search(q, k)
        _check(D, I, corpus, q, k, False, True)
        assert ix.debug_counter(15) == 3
        info = ix.exchange_info()
        assert info["exchange"] == "peer-copy (RCCL failed)" and "injected" in info["rccl_error"]
        assert info["devices"] == [0] and info["peer_access"] == [[-1]]
        D, I = ix.search(q, k)                        # and it stays usable
        _check(D, I, corpus, q, k, False, True)
    finally:
        ix.close()


@pytest.mark.parametrize("nq,k,dtype", [(1, 50, "f32"), (5, 1000, "f32"), (64, 100, "f16")])
def test_enqueue_workers_give_the_same_answer(nq, k, dtype):
    """One host thread per shard queues that shard's work (default on distinct devices; forced on
    here for shards that share the GPU): same results, every call counted by the workers."""
    import torch

    n, d = 140_000, 128
    corpus, q = H.int_corpus(65, n, d), H.int_corpus(66, nq, d)
    ix = FlatIPIndex.from_array(corpus, dtype=dtype, devices=[0, 0, 0, 0])
    try:
        assert ix.exchange_info()["enqueue_workers"] is False   # shards share a device: off
        ix.debug_option(11, 1)
        assert ix.exchange_info()["enqueue_workers"] is True
        for _ in range(3):
            D, I = ix.search(q, k)
            _check(D, I, corpus, q, k, dtype == "f16", True)
        tq = torch.from_numpy(q).cuda(0)
        outs = [ix.search_device(tq, k, asynchronous=True) for _ in range(5)]
        ix.check()
        for s_, i_ in outs:
            _check(s_.cpu().numpy(), i_.cpu().numpy(), corpus, q, k, dtype == "f16", True)
        assert ix.debug_counter(19) == 8
        ix.debug_option(11, 0)
        D, I = ix.search(q, k)
        _check(D, I, corpus, q, k, dtype == "f16", True)
        assert ix.debug_counter(19) == 8
    finally:
        ix.close()


def test_replicated_handle_round_robin_and_concurrent_callers():
    """ls_create_replicated: every device holds the whole corpus, synchronous searches are dealt
    round-robin to the replicas (three replicas rehearsed on this GPU); results are those of a plain
    index, add() reaches every replica, threads calling concurrently are served in parallel."""
    import threading

    n, d, k = 30_000, 96, 40
    corpus, q = H.int_corpus(81, n, d), H.int_corpus(82, 12, d)
    ix = FlatIPIndex.from_array(corpus, devices=[0, 0, 0], replicate=True)
    try:
        assert [s_[1:] for s_ in ix.shards()] == [(0, n)] * 3      # every replica covers all rows
        assert ix.exchange_info()["exchange"].startswith("none (replicas")
        for i in range(7):
            D, I = ix.search(q[i:i + 1], k)
            _check(D, I, corpus, q[i:i + 1], k, False, True)
        assert ix.debug_counter(21) == 7                          # dealt in turn: 3 + 2 + 2
        more = H.int_corpus(83, 500, d)
        ix.add(more)
        both = np.concatenate([corpus, more])
        assert ix.ntotal == n + 500
        for i in range(3):                                        # one call per replica: all of them grew
            D, I = ix.search(q, k)
            _check(D, I, both, q, k, False, True)
        errs = []

        def worker(j):
            try:
                for _ in range(20):
                    D, I = ix.search(q[j:j + 1], k)
                    _check(D, I, both, q[j:j + 1], k, False, True)
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        ts = [threading.Thread(target=worker, args=(j,)) for j in range(6)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errs, errs[:1]
        assert np.array_equal(ix.host_corpus(), both)
        import torch

        s_, i_ = ix.search_device(torch.from_numpy(q).cuda(0), k)   # device queries: the replica on device_ids[0]
        _check(s_.cpu().numpy(), i_.cpu().numpy(), both, q, k, False, True)
    finally:
        ix.close()


def test_group_calls_leave_the_callers_device_current():
    """The group entry points hop over the shards' devices; the calling thread's current device
    (shared with PyTorch) must be what it was, on success and on error."""
    import torch

    corpus, q = H.int_corpus(67, 9_000, 48), H.int_corpus(68, 2, 48)
    ix = FlatIPIndex.from_array(corpus, devices=[0, 0])
    try:
        before = torch.cuda.current_device()
        ix.search(q, 10)
        ix.add(H.int_corpus(69, 100, 48))
        ix.host_corpus()
        ix.debug_counter(8)
        with pytest.raises(Exception):
            ix.search(q, 5000)                        # k too large
        assert torch.cuda.current_device() == before
        x = torch.zeros(4, device="cuda")
        assert x.device.index == before
    finally:
        ix.close()


@pytest.mark.skipif(native.device_count() < 2, reason="needs >= 2 GPUs (auto-runs on a multi-GPU box)")
@pytest.mark.parametrize("mode", [0, 1])
def test_sharded_handle_over_every_visible_gpu(mode):
    """Real multi-device run: one shard per visible GPU, RCCL all-gather over xGMI (mode 0) or peer
    copies (mode 1); scan path, batched path with async + check, bit-exact integer case."""
    import torch

    G = native.device_count()
    n, d = 40_000 * G, 256
    corpus, q = H.gauss(71, n, d), H.gauss(72, 96, d)
    ix = FlatIPIndex.from_array(corpus, dtype="f16", devices=list(range(G)))
    try:
        ix.debug_option(8, mode)
        assert [s[0] for s in ix.shards()] == list(range(G))
        D, I = ix.search(q[:3], 50)
        # mode 0: RCCL in use, or - if RCCL is broken on this node - the reported copy fallback
        assert ix.debug_counter(15) in ((2, 3) if mode == 0 else (0,)), ix.exchange_info()
        assert ix.exchange_info()["enqueue_workers"] is True and ix.debug_counter(19) >= 1
        _check(D, I, corpus, q[:3], 50, True, False)
        D, I = ix.search(q, 100)
        _check(D, I, corpus, q, 100, True, False)
        tq = torch.from_numpy(q).cuda(0)
        outs = [ix.search_device(tq, 100, asynchronous=True) for _ in range(4)]
        ix.check()
        for s, i in outs:
            _check(s.cpu().numpy(), i.cpu().numpy(), corpus, q, 100, True, False)
    finally:
        ix.close()
    ci, qi = H.int_corpus(73, 10_000 * G + 3, 64), H.int_corpus(74, 5, 64)
    ix = FlatIPIndex.from_array(ci, devices=list(range(G)))
    try:
        ix.debug_option(8, mode)
        D, I = ix.search(qi, 1000)
        _check(D, I, ci, qi, 1000, False, True)
    finally:
        ix.close()


def test_sharded_create_errors():
    lib = native.load()
    h = ctypes.c_void_p()
    x = np.zeros((4, 8), np.float32)
    ids = (ctypes.c_int32 * 2)(0, 99)
    assert lib.ls_create_sharded(ctypes.byref(h), x.ctypes.data, 4, 8, 0, ids, 2) == native.LS_ERR_NO_DEVICE
    assert b"out of range" in lib.ls_last_error()
    assert lib.ls_create_sharded(ctypes.byref(h), x.ctypes.data, 4, 8, 0, ids, 0) == native.LS_ERR_NO_DEVICE
    assert lib.ls_create_sharded(ctypes.byref(h), None, 4, 8, 0, ids, 1) == native.LS_ERR_INVALID_ARG
    ok = (ctypes.c_int32 * 2)(0, 0)
    assert lib.ls_create_sharded(ctypes.byref(h), x.ctypes.data, 4, 8, 7, ok, 2) == native.LS_ERR_INVALID_ARG
    assert lib.ls_create_sharded(ctypes.byref(h), x.ctypes.data, 4, 8, 0, ok, 2) == native.LS_OK
    try:
        assert lib.ls_shard_count(h) == 2 and lib.ls_device(h) == 0 and lib.ls_ntotal(h) == 4
        d_dst = ctypes.c_void_p(8)
        assert lib.ls_export_flags(h, d_dst, 1, None) == native.LS_ERR_INVALID_ARG
        assert lib.ls_shard_info(h, 2, None, None, None) == native.LS_ERR_INVALID_ARG
    finally:
        lib.ls_destroy(h)
