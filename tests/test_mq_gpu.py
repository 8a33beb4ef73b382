"""Small batches on an fp32 index (csrc/ls_mq.hip): 2..32 queries share one corpus pass on the f32 matrix
cores (17..32: two 16-column B blocks per A operand, round 6). The kernel is built to reproduce the single-query scan kernel's summation order, so the bar is not a
tolerance: scores AND indices are `array_equal` to (a) the same queries served one by one and (b) the CPU
oracle in the documented "scan" order (oracle.compare_kernel_order), in addition to the usual 1e-5 /
near-tie check against the strict oracle. Reference call being replaced: `index.search(x, k)`,
src/lean_explore/search/engine.py:250 (issued concurrently by several MCP clients, mcp/server.py:147-151)."""

import numpy as np
import pytest

from lean_explore_amd.index import FlatIPIndex
from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu


def one_by_one(ix, q, k, normalize):
    outs = [ix.search(q[j:j + 1], k, normalize=normalize) for j in range(q.shape[0])]
    return np.concatenate([o[0] for o in outs]), np.concatenate([o[1] for o in outs])


# (k against the shard size decides how many keys a lane keeps - 3, 5 or 8, ls_mq_lane_keys - or whether the
# VALU scan groups take the call: same bits either way)
@pytest.mark.parametrize("d,k,normalize", [(384, 50, True), (1024, 1000, True), (100, 10, False), (64, 100, False),
                                           (768, 200, True), (200, 50, False), (512, 50, True), (36, 7, False)])
def test_mq_equals_single_queries_and_the_scan_order_oracle(d, k, normalize):
    n = 40_000 if d <= 512 else 24_000
    c = H.gauss(100 + d, n, d)
    q = H.gauss(200 + d, 32, d, normalize=not normalize) * (1.0 if not normalize else 3.0)
    ix = FlatIPIndex.from_array(c)
    D1, I1 = one_by_one(ix, q, k, normalize)
    qn = oracle.c_normalize_l2(q) if normalize else q
    for nq in (2, 3, 5, 8, 13, 16, 17, 20, 24, 31, 32):  # (33 and up: the batched f32 MFMA path, tests/test_batched_gpu.py)
        before = ix.debug_counter(23)
        D, I = ix.search(q[:nq], k, normalize=normalize)
        if k < 1000:  # (k = 1000 of 24 k rows may or may not fit the lanes' key lists: same bits either way)
            assert ix.debug_counter(23) > before, "the small batch did not take the f32 MFMA kernel"
            # 17..32 queries are ONE exact pass (two B blocks), not a speculative batch
            assert ix.debug_counter(23) == before + 1 and ix.debug_counter(10) == 1, (nq, ix.debug_counter(10))
        assert np.array_equal(D, D1[:nq]) and np.array_equal(I, I1[:nq]), (d, k, nq)
        rep = oracle.compare_kernel_order(D, I, c, qn[:nq], k, orders=("scan",))
        assert rep["kernel_order_mismatches"] == 0
        Dr, Ir = oracle.c_search(c, qn[:nq], k)
        _, _, S = oracle.np_search(c, qn[:nq], k)
        rep = oracle.compare_topk(D, I, Dr, Ir, S, score_tol=1e-5)
        assert rep["recall"] == 1.0, rep
    ix.close()


def test_mq_full_size_config2_shapes():
    """N = 200 k: d = 384 k = 50 (config 2's shape, 16 queries per pass) and d = 1024 k = 1000 (the
    reference's call shape, engine.py:538): bit-identical to the scan-order oracle."""
    for d, k in ((384, 50), (1024, 1000)):
        c = H.gauss(1234, 200_000, d)
        q = H.gauss(5678, 16, d)
        ix = FlatIPIndex.from_array(c)
        D, I = ix.search(q, k)
        assert ix.debug_counter(23) >= 1
        rep = oracle.compare_kernel_order(D, I, c, q, k, orders=("scan",))
        D8, I8 = ix.search(q[:8], k)
        assert np.array_equal(D8, D[:8]) and np.array_equal(I8, I[:8])
        # 32 queries, ONE pass (two 16-column blocks): the same bits for the queries both calls share
        q32 = np.concatenate([q, H.gauss(91011, 16, d)])
        before = ix.debug_counter(23)
        D32, I32 = ix.search(q32, k)
        assert ix.debug_counter(23) == before + 1, "32 queries must be one ls_mq launch"
        assert np.array_equal(D32[:16], D) and np.array_equal(I32[:16], I)
        oracle.compare_kernel_order(D32, I32, c, q32, k, orders=("scan",))
        ix.debug_option(22, 0)   # 16 columns per pass: two launches, same bits
        D32b, I32b = ix.search(q32[:23], k)
        assert np.array_equal(D32b, D32[:23]) and np.array_equal(I32b, I32[:23])
        ix.debug_option(22, 1)
        ix.debug_option(16, 0)   # the VALU scan groups of 8 / 4 / 1: same bits
        Dv, Iv = ix.search(q, k)
        assert np.array_equal(Dv, D) and np.array_equal(Iv, I)
        if k <= 50:  # (at k = 1000 a lane holding 5 of a top-1000 is a 3e-3 event per query: exact either way)
            assert ix.debug_counter(0) == 0, "random data must stay on the selection's fast path"
        ix.close()
        print("mq full size", d, k, rep)


@pytest.mark.parametrize("n", [4096, 4097, 5000, 8191, 16_400, 33_333])
def test_mq_ragged_shards_and_padding(n):
    c = H.gauss(n, n, 384)
    q = H.gauss(n + 1, 7, 384)
    ix = FlatIPIndex.from_array(c, base=10_000_000_000)
    for k in (1, 50, 300):
        D, I = ix.search(q, k)
        oracle.compare_kernel_order(D, I, c, q, k, base=10_000_000_000, orders=("scan",))
    ix.close()


def test_mq_integer_corpus_ties_and_clusters():
    """Exact arithmetic with thousands of ties, a clustered (sorted) corpus whose best rows are adjacent
    - the per-lane key lists overflow, the proof fails, the rescue over S must still give the exact answer -
    and NaN / inf rows."""
    c = H.int_corpus(7, 60_000, 128)
    q = H.int_corpus(8, 9, 128)
    ix = FlatIPIndex.from_array(c)
    for k in (50, 1000):
        D, I = ix.search(q, k)
        Dr, Ir = oracle.c_search(c, q, k)
        assert np.array_equal(D, Dr) and np.array_equal(I, Ir)
    ix.close()
    c = H.gauss(3, 60_000, 384)
    q = H.gauss(4, 6, 384)
    order = np.argsort(c @ q[0])
    c = np.ascontiguousarray(c[order])
    ix = FlatIPIndex.from_array(c)
    D, I = ix.search(q, 200)
    oracle.compare_kernel_order(D, I, c, q, 200, orders=("scan",))
    # (host call: the launch wrote no score vectors, the unproven queries were served again on the scan kernel -
    # counter 25 - whose own selection may or may not need its slow path - counter 0)
    assert ix.debug_counter(0) + ix.debug_counter(25) >= 1, "the sorted corpus should have forced a repair"
    ix.close()
    c2 = H.gauss(8, 10_000, 64)
    c2[10, 0] = np.nan
    c2[11, 0] = -np.inf
    c2[12, 0] = np.inf
    ix = FlatIPIndex.from_array(c2)
    qq = np.ones((3, 64), np.float32)
    D, I = ix.search(qq, 100)
    Dr, Ir = oracle.c_search(c2, qq, 100)
    assert np.array_equal(I, Ir) and 10 not in I[0] and 11 not in I[0] and I[0, 0] == 12
    ix.close()


def test_mq_device_api_pipelined_and_async():
    import torch

    c = H.gauss(41, 50_000, 384)
    q = H.gauss(42, 100, 384)
    ix = FlatIPIndex.from_array(c)
    tq = torch.from_numpy(q).cuda()
    outs = []
    sizes = [2, 16, 5, 1, 9, 7, 32, 17, 3, 8]
    at = 0
    for m in sizes:
        outs.append((at, m, ix.search_device(tq[at:at + m], 50, pipeline=True)))
        at += m
    ix.check()
    for at, m, (Dt, It) in outs:
        oracle.compare_kernel_order(Dt.cpu().numpy(), It.cpu().numpy(), c, q[at:at + m], 50, orders=("scan",))
    Dt, It = ix.search_device(tq[:12], 100, asynchronous=True)
    ix.check()
    oracle.compare_kernel_order(Dt.cpu().numpy(), It.cpu().numpy(), c, q[:12], 100, orders=("scan",))
    # LS_FLAG_ASYNC alone keeps the score vectors: a 32-query pass needs 32 of them (grown on demand)
    Dt, It = ix.search_device(tq[:32], 100, asynchronous=True)
    torch.cuda.synchronize()
    oracle.compare_kernel_order(Dt.cpu().numpy(), It.cpu().numpy(), c, q[:32], 100, orders=("scan",))
    ix.close()


@pytest.mark.parametrize("d,k", [(384, 100), (1024, 100), (768, 300)])
def test_wide_pass_repairs_and_retries(d, k):
    """17..32 queries per pass (two B blocks; 4 KB rows: one workgroup per CU, the selection workgroups run two
    jobs each): a clustered corpus and k' = 1 force every query through the retry / repair paths - host call
    (completion words, the stand-alone finalize or a second serve), pipelined device call (repair at ls_check),
    LS_FLAG_ASYNC (score vectors kept: the rescue sweeps S). Same bits as one query at a time."""
    import torch

    c = H.gauss(3, 30_000, d)
    q = H.gauss(4, 29, d)
    c = np.ascontiguousarray(c[np.argsort(c @ q[0])])
    ix = FlatIPIndex.from_array(c)
    D1, I1 = one_by_one(ix, q, k, False)
    oracle.compare_kernel_order(D1, I1, c, q, k, orders=("scan",))
    D, I = ix.search(q, k)
    assert np.array_equal(D, D1) and np.array_equal(I, I1)
    ix.debug_option(0, 1)  # k' = 1: nothing can be proven from the workgroups' keys
    before = ix.debug_counter(25)
    D, I = ix.search(q, k)
    assert ix.debug_counter(25) >= before + 20
    assert np.array_equal(D, D1) and np.array_equal(I, I1)
    tq = torch.from_numpy(q).cuda()
    Dt, It = ix.search_device(tq, k, pipeline=True)
    Dt2, It2 = ix.search_device(tq[:19], k, pipeline=True)
    ix.check()
    assert np.array_equal(Dt.cpu().numpy(), D1) and np.array_equal(It.cpu().numpy(), I1)
    assert np.array_equal(Dt2.cpu().numpy(), D1[:19]) and np.array_equal(It2.cpu().numpy(), I1[:19])
    Dt, It = ix.search_device(tq, k, asynchronous=True)
    torch.cuda.synchronize()
    assert np.array_equal(Dt.cpu().numpy(), D1) and np.array_equal(It.cpu().numpy(), I1)
    ix.debug_option(19, 0)  # host calls keep their score vectors: the stand-alone finalize rescues from S
    D, I = ix.search(q, k)
    assert np.array_equal(D, D1) and np.array_equal(I, I1)
    ix.close()


def test_one_query_per_launch_with_many_groups_keeps_its_retries_straight():
    """with one query per launch (debug option 6 = 0) a synchronous call of 3+ queries used more
    scratch generations than exist while same-launch retry jobs still pointed at them; a retry (k' = 1 on a
    clustered corpus forces one) then read another query's score vector. Such calls now take the
    selection's own launch; the answers must be those of the oracle."""
    c = H.gauss(3, 60_000, 384)
    q = H.gauss(4, 6, 384)
    order = np.argsort(c @ q[0])
    c = np.ascontiguousarray(c[order])
    ix = FlatIPIndex.from_array(c)
    ix.debug_option(6, 0)
    ix.debug_option(0, 1)
    D, I = ix.search(q, 100)
    oracle.compare_kernel_order(D, I, c, q, 100, orders=("scan",))
    ix.close()


def test_host_calls_without_score_vectors_are_served_again_when_unproven():
    """Synchronous host calls (the reference's call, engine.py:250, several callers combined): the ls_mq launch
    writes no score vectors (debug option 19); a query whose workgroup keys cannot be proven complete - a
    clustered corpus, k' forced to 1 - is served again, alone, on the scan kernel. Same bits either way."""
    c = H.gauss(3, 60_000, 384)
    q = H.gauss(4, 9, 384)
    c = np.ascontiguousarray(c[np.argsort(c @ q[0])])
    ix = FlatIPIndex.from_array(c)
    D, I = ix.search(q, 100)
    assert ix.debug_counter(25) >= 1, "the clustered query was not served again"
    oracle.compare_kernel_order(D, I, c, q, 100, orders=("scan",))
    ix.debug_option(0, 1)  # k' = 1: every query needs the repair
    before = ix.debug_counter(25)
    D1, I1 = ix.search(q, 100)
    assert ix.debug_counter(25) >= before + 5
    assert np.array_equal(D1, D) and np.array_equal(I1, I)
    ix.debug_option(19, 0)  # with score vectors: the stand-alone selection repairs from S
    before = ix.debug_counter(25)
    D2, I2 = ix.search(q, 100)
    assert ix.debug_counter(25) == before
    assert np.array_equal(D2, D) and np.array_equal(I2, I)
    ix.close()


def test_device_calls_without_score_vectors_are_repaired_at_check():
    """Device-output calls: pipelined results are final after ls_check, synchronous ones when the call returns -
    their ls_mq launches write no score vectors either, keep their raw queries, and an unproven query (clustered
    corpus) is served again in place by the scan kernel. LS_FLAG_ASYNC alone promises stream order: it keeps S."""
    import torch

    c = H.gauss(3, 60_000, 384)
    q = H.gauss(4, 12, 384)
    c = np.ascontiguousarray(c[np.argsort(c @ q[0])])
    ix = FlatIPIndex.from_array(c)
    tq = torch.from_numpy(q).cuda()
    outs = [ix.search_device(tq[:m], 100, pipeline=True) for m in (12, 2, 7)]
    tq2 = tq.clone()
    torch.cuda.synchronize()
    tq.zero_()  # (the passes have consumed the queries: the repair works from the launches' own copies)
    torch.cuda.synchronize()
    ix.check()
    assert ix.debug_counter(25) >= 3, "the clustered query was not served again"
    for m, (Dt, It) in zip((12, 2, 7), outs):
        oracle.compare_kernel_order(Dt.cpu().numpy(), It.cpu().numpy(), c, q[:m], 100, orders=("scan",))
    before = ix.debug_counter(25)
    Dt, It = ix.search_device(tq2[:9], 100)  # synchronous: repaired before it returns
    assert ix.debug_counter(25) > before
    oracle.compare_kernel_order(Dt.cpu().numpy(), It.cpu().numpy(), c, q[:9], 100, orders=("scan",))
    before = ix.debug_counter(25)
    Dt, It = ix.search_device(tq2[:9], 100, asynchronous=True)
    torch.cuda.synchronize()
    assert ix.debug_counter(25) == before and ix.debug_counter(0) >= 1
    oracle.compare_kernel_order(Dt.cpu().numpy(), It.cpu().numpy(), c, q[:9], 100, orders=("scan",))
    # more pipelined launches than kept-query slots between two checks: the ring forces a repair
    ix.debug_option(0, 1)
    outs = [ix.search_device(tq2[:3], 50, pipeline=True) for _ in range(300)]
    ix.check()
    D0, I0 = outs[0][0].cpu().numpy(), outs[0][1].cpu().numpy()
    oracle.compare_kernel_order(D0, I0, c, q[:3], 50, orders=("scan",))
    for Dt, It in outs[1:]:
        assert np.array_equal(Dt.cpu().numpy(), D0) and np.array_equal(It.cpu().numpy(), I0)
    ix.close()


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_single_query_device_calls_without_score_vector_are_repaired(dtype):
    """The headline's loop - pipelined single-query device calls - writes no score vector either (the riding
    selection workgroup keeps the raw query); with k' forced to 1 every query needs the repair at ls_check.
    Option 19 = 2 keeps the vector: same results."""
    import torch

    c = H.gauss(31, 50_000, 384)
    q = H.gauss(32, 6, 384)
    ix = FlatIPIndex.from_array(c, dtype=dtype)
    tq = torch.from_numpy(q).cuda()
    ref = [ix.search_device(tq[j:j + 1], 64, asynchronous=True) for j in range(6)]  # (keeps S: exact in stream order)
    torch.cuda.synchronize()
    ix.debug_option(0, 1)
    before = ix.debug_counter(25)
    outs = [ix.search_device(tq[j:j + 1], 64, pipeline=True) for j in range(6)]
    ix.check()
    assert ix.debug_counter(25) >= before + 4
    for (Dr, Ir), (Dt, It) in zip(ref, outs):
        assert torch.equal(Dr, Dt) and torch.equal(Ir, It)
    Ds, Is = ix.search_device(tq[2:3], 64)  # synchronous
    assert torch.equal(Ds, ref[2][0]) and torch.equal(Is, ref[2][1])
    ix.debug_option(19, 2)
    before = ix.debug_counter(25)
    outs = [ix.search_device(tq[j:j + 1], 64, pipeline=True) for j in range(6)]
    ix.check()
    assert ix.debug_counter(25) == before
    for (Dr, Ir), (Dt, It) in zip(ref, outs):
        assert torch.equal(Dr, Dt) and torch.equal(Ir, It)
    if dtype == "f32":
        oracle.compare_kernel_order(ref[0][0].cpu().numpy(), ref[0][1].cpu().numpy(), c, q[:1], 64, orders=("scan",))
    ix.close()


def test_inorder_flag_keeps_pipelined_results_exact_before_check():
    """LS_FLAG_INORDER (what the torchrun exchange passes): pipelined scan-path launches keep their score vectors,
    so an unproven query is rescued inside the stream - nothing is left for ls_check to repair."""
    import torch

    c = H.gauss(3, 60_000, 384)
    q = H.gauss(4, 6, 384)
    c = np.ascontiguousarray(c[np.argsort(c @ q[0])])
    ix = FlatIPIndex.from_array(c)
    tq = torch.from_numpy(q).cuda()
    before = ix.debug_counter(25)
    outs = [ix.search_device(tq[:m], 100, pipeline=True, inorder=True) for m in (6, 1, 3, 1)]
    ix.check()
    assert ix.debug_counter(25) == before and ix.debug_counter(0) >= 1
    for m, (Dt, It) in zip((6, 1, 3, 1), outs):
        oracle.compare_kernel_order(Dt.cpu().numpy(), It.cpu().numpy(), c, q[:m], 100, orders=("scan",))
    ix.close()


def test_set_base_repairs_pending_queries_under_the_old_base():
    """A pipelined launch without score vectors that raised repair words, then ls_set_base before ls_check: the
    queries are served again under the base they were submitted with (ls_set_base repairs first)."""
    import torch

    from lean_explore_amd import native

    c = H.gauss(31, 50_000, 384)
    q = H.gauss(32, 5, 384)
    ix = FlatIPIndex.from_array(c)
    tq = torch.from_numpy(q).cuda()
    D0, I0 = ix.search(q, 64)
    ix.debug_option(0, 1)  # k' = 1: every query needs the repair
    Dt, It = ix.search_device(tq, 64, pipeline=True)
    native.check(native.load().ls_set_base(ix._handle, 1_000_000))
    ix.check()
    assert np.array_equal(Dt.cpu().numpy(), D0) and np.array_equal(It.cpu().numpy(), I0)
    D1, I1 = ix.search(q, 64)
    assert np.array_equal(D1, D0) and np.array_equal(I1, I0 + 1_000_000)
    ix.close()
