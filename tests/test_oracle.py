"""CPU tests of the oracle itself: pinned against the reference's known-answer test and the
committed golden vectors, and cross-checked C <-> numpy twin. No GPU needed."""

import numpy as np
import pytest

from oracle import oracle
from tests import helpers as H

NEG = oracle.NEG_FLT_MAX


def test_reference_known_answer_e0():
    """reference tests/extract/index_test.py:186-205: row 0 = e0, query e0, k=1 -> label 0.
    True for any seed (all other scores are r[0] < 1); try several, as the reference is unseeded."""
    for seed in (0, 1, 2, 20240611):
        emb, q = H.kat_inputs(seed)
        D, I = oracle.c_search(emb, q, 1)
        assert I[0, 0] == 0 and D[0, 0] == 1.0
        D2, I2, _ = oracle.np_search(emb, q, 1)
        assert I2[0, 0] == 0


def test_reference_structural_shapes():
    """reference tests/conftest.py:168-181 documents the return shape ([[..]], [[..]])."""
    c = H.gauss(1, 300, 768)
    D, I = oracle.c_search(c, c[:1], 3)
    assert D.shape == (1, 3) and I.shape == (1, 3)
    assert D.dtype == np.float32 and I.dtype == np.int64
    assert I[0, 0] == 0  # a normalised row is its own nearest neighbour


def test_golden_dense_vectors():
    meta, arr = H.load_golden()
    for name, m in meta.items():
        if name in ("kat", "tie"):
            continue
        c = H.gauss(m["corpus_seed"], m["n"], m["d"])
        q = H.gauss(m["query_seed"], m["nq"], m["d"])
        assert H.sha(c) == m["corpus_sha"] and H.sha(q) == m["query_sha"], "RNG drift"
        D, I = oracle.c_search(c, q, m["k"], f16=m["f16"])
        _, _, S = oracle.np_search(c, q, m["k"], f16=m["f16"])
        rep = oracle.compare_topk(D, I, arr[f"{name}__D"], arr[f"{name}__I"], S, score_tol=2e-6)
        assert rep["recall"] == 1.0, (name, rep)


def test_golden_kat_and_tie():
    meta, arr = H.load_golden()
    emb, q = H.kat_inputs(meta["kat"]["seed"])
    assert H.sha(emb) == meta["kat"]["corpus_sha"]
    D, I = oracle.c_search(emb, q, 1)
    assert np.array_equal(I, arr["kat__I"]) and np.array_equal(D, arr["kat__D"])
    c = H.int_corpus(99, 4096, 64)
    qq = np.random.default_rng(99)
    _ = qq.integers(-3, 4, size=(4096, 64))
    qi = qq.integers(-3, 4, size=(3, 64)).astype(np.float32)
    assert H.sha(c) == meta["tie"]["corpus_sha"] and H.sha(qi) == meta["tie"]["query_sha"]
    D, I = oracle.c_search(c, qi, 100)
    # integer data: exact arithmetic -> bit-exact scores and the (score desc, row asc) order
    assert np.array_equal(I, arr["tie__I"]) and np.array_equal(D, arr["tie__D"])
    # ties really occur and are broken by ascending row
    eq = D[:, 1:] == D[:, :-1]
    assert eq.any()
    assert (I[:, 1:][eq] > I[:, :-1][eq]).all()


@pytest.mark.parametrize("f16", [False, True])
def test_c_matches_numpy_twin(f16):
    c = H.gauss(11, 3000, 384)
    q = H.gauss(12, 5, 384)
    D, I = oracle.c_search(c, q, 64, f16=f16)
    Dn, In, S = oracle.np_search(c, q, 64, f16=f16)
    rep = oracle.compare_topk(D, I, Dn, In, S, score_tol=2e-6)
    assert rep["recall"] == 1.0
    assert (np.diff(D, axis=1) <= 0).all()  # best first


def test_f16_rounding_matches_numpy():
    x = np.concatenate([H.gauss(3, 50, 64).ravel() * s for s in (1, 1e-3, 1e-6, 1e-8, 7e4, 1e6)])
    x = np.concatenate([x, np.array([0.0, -0.0, 65504.0, 65520.0, 6e-8, 5.96e-8, 2.98e-8],
                                    np.float32)]).astype(np.float32)
    got = oracle.c_round_f16(x)
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).astype(np.float32)
    assert np.array_equal(got, want)


def test_padding_when_k_exceeds_n():
    c = H.gauss(5, 7, 32)
    q = H.gauss(6, 2, 32)
    D, I = oracle.c_search(c, q, 10)
    assert (I[:, 7:] == -1).all() and (D[:, 7:] == NEG).all()
    assert sorted(I[0, :7]) == list(range(7))
    D0, I0 = oracle.c_search(np.zeros((0, 32), np.float32), q, 4)
    assert (I0 == -1).all() and (D0 == NEG).all()


def test_duplicate_rows_tie_break_ascending_row():
    c = H.gauss(7, 100, 64)
    c[40] = c[3]
    c[77] = c[3]
    D, I = oracle.c_search(c, c[3:4], 5)
    assert list(I[0, :3]) == [3, 40, 77]
    assert D[0, 0] == D[0, 1] == D[0, 2]


def test_negative_scores_zero_query_nan_rows():
    c = H.gauss(8, 50, 16)
    D, I = oracle.c_search(-np.abs(c), np.abs(c[:1]), 50)
    assert (D < 0).all() and (np.diff(D, axis=1) <= 0).all()
    Dz, Iz = oracle.c_search(c, np.zeros((1, 16), np.float32), 5)
    assert (Dz == 0).all() and list(Iz[0]) == [0, 1, 2, 3, 4]  # all tie at 0 -> lowest rows
    c2 = c.copy()
    c2[10, 0] = np.nan
    c2[11, 0] = -np.inf
    Dn, In = oracle.c_search(c2, np.ones((1, 16), np.float32), 50)
    assert 10 not in In[0] and 11 not in In[0]
    assert (In[0, 48:] == -1).all() and (Dn[0, 48:] == NEG).all()


def test_normalize_l2_semantics():
    x = H.gauss(9, 6, 384, normalize=False) * 3.0
    x[2] = 0.0
    y = oracle.c_normalize_l2(x)
    assert np.allclose(np.linalg.norm(y[[0, 1, 3, 4, 5]], axis=1), 1.0, atol=1e-6)
    assert (y[2] == 0).all()  # zero rows untouched (faiss fvec_renorm_L2)
    assert np.allclose(y, oracle.np_normalize_l2(x), atol=1e-6)
    assert np.array_equal(x[0], H.gauss(9, 6, 384, normalize=False)[0] * 3.0)  # input not mutated


def test_merge_equals_unsharded():
    c = H.int_corpus(21, 1000, 32)
    q = H.int_corpus(22, 4, 32)
    k = 30
    Dref, Iref = oracle.c_search(c, q, k)
    for g in (1, 2, 3, 8):
        bounds = [(i * 1000) // g for i in range(g + 1)]
        parts = [oracle.c_search(c[a:b], q, k, base=a) for a, b in zip(bounds[:-1], bounds[1:])]
        Dm, Im = oracle.c_merge(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]))
        assert np.array_equal(Dm, Dref) and np.array_equal(Im, Iref)


def test_fast_variant_agrees():
    c = H.gauss(31, 2000, 384)
    q = H.gauss(32, 3, 384)
    D, I = oracle.c_search(c, q, 50)
    Df, If = oracle.c_search(c, q, 50, fast=True)
    _, _, S = oracle.np_search(c, q, 50)
    oracle.compare_topk(Df, If, D, I, S)


def test_third_checker_torch_cpu_topk():
    """An independent third implementation (torch CPU: fp32 sgemm + topk) agrees with the C oracle
    and its float64 numpy twin. It cannot pin parity to FAISS either; it removes "the two
    restatements share an author's bug" as an objection."""
    import torch

    c, q = H.gauss(1234, 20_000, 384), H.gauss(5678, 7, 384)
    k = 50
    Dc, Ic = oracle.c_search(c, q, k)
    _, _, S = oracle.np_search(c, q, k)
    ts, ti = torch.topk(torch.from_numpy(q) @ torch.from_numpy(c).T, k, dim=1)
    rep = oracle.compare_topk(ts.numpy(), ti.numpy(), Dc, Ic, S, score_tol=1e-5, tie_eps=2e-6)
    assert rep["recall"] == 1.0, rep
    # fp16 storage semantics: both operands rounded, fp32 accumulate
    Dh, Ih = oracle.c_search(c, q, k, f16=True)
    _, _, Sh = oracle.np_search(c, q, k, f16=True)
    th, tih = torch.topk(torch.from_numpy(q).half().float() @ torch.from_numpy(c).half().float().T,
                         k, dim=1)
    assert oracle.compare_topk(th.numpy(), tih.numpy(), Dh, Ih, Sh)["recall"] == 1.0


def test_normalize_order_is_the_documented_one():
    """oracle_normalize_l2 = 64 interleaved fma partial sums + xor tree 32..1 (the order
    ls_wave_sumsq uses on the GPU), restated here in numpy float32."""
    x = (H.gauss(9, 4, 777, normalize=False) * 1.3).astype(np.float32)
    got = oracle.c_normalize_l2(x)
    for r in range(x.shape[0]):
        p = np.zeros(64, dtype=np.float64)
        for j in range(x.shape[1]):  # fma(x, x, p): exact product, one rounding
            p[j & 63] = np.float32(np.float64(x[r, j]) * np.float64(x[r, j]) + p[j & 63])
        p = p.astype(np.float32)
        o = 32
        while o >= 1:
            p[:o] = p[:o] + p[o:2 * o]
            o >>= 1
        inv = np.float32(1.0) / np.sqrt(p[0], dtype=np.float32)
        assert np.array_equal(got[r], x[r] * inv)


@pytest.mark.parametrize("order", ["scan", "fma"])
def test_kernel_order_modes_against_the_twin_and_on_exact_data(order):
    """The oracle's kernel-order modes (flat_ip_ref.c ORDER_SCAN / ORDER_FMA: the fp32 HIP kernels' own
    documented summation orders) are still the same inner product: within 1e-6 of the float64 twin on
    continuous data for every row geometry, near-tie-equivalent to the strict order, bit-identical to it on
    integer data (every order is exact there), and they pass the reference's known-answer test."""
    emb, q = H.kat_inputs(5)
    D, I = oracle.c_search(emb, q, 1, order=order)
    assert I[0, 0] == 0 and D[0, 0] == 1.0
    for d in (17, 64, 100, 200, 384, 500, 768, 1024):
        c = H.gauss(d, 3000, d)
        qq = H.gauss(d + 1, 3, d)
        D, I = oracle.c_search(c, qq, 40, order=order)
        Ds, Is = oracle.c_search(c, qq, 40)
        _, _, S = oracle.np_search(c, qq, 40)
        rep = oracle.compare_topk(D, I, Ds, Is, S, score_tol=1e-6)
        assert rep["recall"] == 1.0, (d, rep)
        got = oracle.c_scores(c, qq[0], order=order)
        assert np.abs(got.astype(np.float64) - S[0]).max() < 1e-6
        ci = H.int_corpus(d, 2000, d)
        qi = H.int_corpus(d + 1, 2, d)
        Di, Ii = oracle.c_search(ci, qi, 100, order=order)
        Dx, Ix = oracle.c_search(ci, qi, 100)
        assert np.array_equal(Di, Dx) and np.array_equal(Ii, Ix), d


def test_scan_order_is_the_documented_lane_tree():
    """ORDER_SCAN spelled out independently in numpy float32 for one geometry (d = 384 -> 96 chunks,
    L = 32 lanes x V = 3 chunks): per-lane fused chains over chunks sub, sub + 32, sub + 64, then the
    balanced tree - csrc/ls_scan.hip:66-77,129-145. math.fma is not available on Python 3.10, so the
    fused multiply-add is emulated exactly in float64 (a product of two floats is exact in float64; the
    sum of that product and a float rounds once to float64 and then to float32 - double rounding can
    differ from a true fma only when the float64 sum lands exactly on a float32 tie, which the assert
    below tolerates for at most a handful of rows)."""
    assert oracle.geom_f32(384) == (32, 3) and oracle.geom_f32(1024) == (64, 4) and oracle.geom_f32(100) == (16, 2)
    c = H.gauss(3, 500, 384)
    q = H.gauss(4, 1, 384)[0]
    want = oracle.c_scores(c, q, order="scan")
    L, V = 32, 3
    p = np.zeros((500, L), np.float32)
    for v in range(V):
        for j in range(4):
            e = 4 * (np.arange(L) + L * v) + j
            p = (c[:, e].astype(np.float64) * q[e].astype(np.float64) + p.astype(np.float64)).astype(np.float32)
    o = 1
    while o < L:
        p = (p[:, 0::2] + p[:, 1::2]).astype(np.float32)
        o *= 2
    assert (p[:, 0] != want).sum() <= 2, (p[:, 0] != want).sum()


def test_compare_kernel_order_rejects_the_wrong_order():
    c = H.gauss(1, 20_000, 384)
    q = H.gauss(2, 3, 384)
    D, I = oracle.c_search(c, q, 50, order="scan")
    assert oracle.compare_kernel_order(D, I, c, q, 50)["kernel_order_queries"] == {"scan": 3}
    Ds, Is = oracle.c_search(c, q, 50)  # the strict order is NOT the kernels' order: last-bit differences
    with pytest.raises(AssertionError):
        oracle.compare_kernel_order(Ds, Is, c, q, 50)
