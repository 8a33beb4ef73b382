"""TEST INFRASTRUCTURE ONLY — Python face of the CPU oracle.

Two independent restatements of the reference's dense-retrieval arithmetic
(reference src/lean_explore/search/engine.py:238-258; see flat_ip_ref.c for the full header):

* ``c_*``  : ctypes bindings of oracle/flat_ip_ref.c (strict left-to-right fp32).
* ``np_*`` : a numpy twin (float64 accumulate, ``np.lexsort``) used to cross-check the C file
  and as the "ground truth" when a near-tie makes fp32 summation order matter.
* ``order="scan" / "fma"`` of the C file + ``compare_kernel_order``: the fp32 kernels' own documented
  summation orders restated on the CPU, so that fp32 results are checked BIT FOR BIT (scores and
  indices, no near-tie excuse) in addition to the 1e-5 / near-tie check against the strict order.

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may import this
module. The product package (lean-explore_amd/) never does.

PARITY UNPINNED beyond the reference's one known-answer test
(reference tests/extract/index_test.py:186-205) and its structural asserts: faiss itself is
not available, so summation order and tie-break are fixed by definition (flat_ip_ref.c header).
"""

from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_BUILD = _HERE / "_build"
NEG_FLT_MAX = np.float32(-3.4028234663852886e38)

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force: bool = False) -> None:
    """Compile flat_ip_ref.c with gcc (strict + fast variants)."""
    if force or not (_BUILD / "liboracle.so").exists() or not (_BUILD / "liboracle_fast.so").exists():
        subprocess.run(["make", "-C", str(_HERE), "-s"], check=True)


def _load(name: str) -> ctypes.CDLL:
    path = _BUILD / name
    if not path.exists():
        build()
    lib = ctypes.CDLL(str(path))
    lib.oracle_flat_ip_topk.restype = ctypes.c_int
    lib.oracle_flat_ip_topk.argtypes = [
        _f32p, ctypes.c_int64, ctypes.c_int32, _f32p, ctypes.c_int64, ctypes.c_int32,
        ctypes.c_int64, _f32p, _i64p,
    ]
    lib.oracle_flat_ip_topk_order.restype = ctypes.c_int
    lib.oracle_flat_ip_topk_order.argtypes = [
        _f32p, ctypes.c_int64, ctypes.c_int32, _f32p, ctypes.c_int64, ctypes.c_int32,
        ctypes.c_int64, ctypes.c_int32, _f32p, _i64p,
    ]
    lib.oracle_scores_order.restype = ctypes.c_int
    lib.oracle_scores_order.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int32, _f32p, ctypes.c_int32, _f32p]
    lib.oracle_geom_f32.restype = ctypes.c_int
    lib.oracle_geom_f32.argtypes = [ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
    lib.oracle_merge_topk.restype = ctypes.c_int
    lib.oracle_merge_topk.argtypes = [
        _f32p, _i64p, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, _f32p, _i64p,
    ]
    lib.oracle_normalize_l2.restype = None
    lib.oracle_normalize_l2.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int32]
    lib.oracle_round_f16.restype = None
    lib.oracle_round_f16.argtypes = [_f32p, ctypes.c_int64]
    lib.oracle_scores.restype = None
    lib.oracle_scores.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int32, _f32p, _f32p]
    lib.oracle_num_threads.restype = ctypes.c_int
    lib.oracle_set_num_threads.argtypes = [ctypes.c_int]
    return lib


_LIBS: dict[str, ctypes.CDLL] = {}


def lib(fast: bool = False) -> ctypes.CDLL:
    name = "liboracle_fast.so" if fast else "liboracle.so"
    if name not in _LIBS:
        _LIBS[name] = _load(name)
    return _LIBS[name]


def _f32(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a: np.ndarray, t):
    return a.ctypes.data_as(t)


# --------------------------------------------------------------------------- C oracle
def c_normalize_l2(x: np.ndarray) -> np.ndarray:
    """Return an L2-normalised copy (faiss.normalize_L2 semantics, zero rows untouched)."""
    y = _f32(x).copy()
    if y.ndim == 1:
        y = y[None, :]
    lib().oracle_normalize_l2(_ptr(y, _f32p), y.shape[0], y.shape[1])
    return y


def c_round_f16(x: np.ndarray) -> np.ndarray:
    y = _f32(x).copy()
    lib().oracle_round_f16(_ptr(y, _f32p), y.size)
    return y


# fp32 summation orders of oracle_flat_ip_topk_order (flat_ip_ref.c):
#   "strict": left to right, one rounding per multiply and per add - the DEFINITION scores are held to
#             (1e-5, BASELINE.json) and what compare_topk's near-tie excuse refers to;
#   "scan"  : the fp32 scan kernels' own order (per-lane fmaf chains over chunks sub, sub+L, .. then the
#             xor tree; csrc/ls_scan.hip, and the f32 MFMA small-batch kernel csrc/ls_mq.hip reproduces
#             it) - every fp32 search of nq < LS_GEMM32_MIN_NQ queries must match it BIT FOR BIT (round 6: and
#             every search of up to 32 queries where ls_mq serves the index - one exact pass of two B blocks);
#   "fma"   : one sequential fmaf chain in increasing k (csrc/ls_gemm32.hip's v_mfma_f32_16x16x4_f32,
#             measured bit-identical to that chain by tools/arith_probe.hip) - fp32 batches of more than 32
#             queries, or of >= LS_GEMM32_MIN_NQ where ls_mq does not serve the (index, k) (a query repaired
#             through the scan path follows "scan"). For 24..32 queries compare_kernel_order admits both.
ORDERS = {"strict": 0, "scan": 1, "fma": 2}
SCAN_PATH_MAX_NQ_F32 = 23  # LS_GEMM32_MIN_NQ - 1 (csrc/ls_common.h)


def kernel_order(nq: int) -> str:
    """The summation order the library uses for an fp32 index and a call of nq queries."""
    return "scan" if nq <= SCAN_PATH_MAX_NQ_F32 else "fma"


def geom_f32(d: int) -> tuple[int, int]:
    L, V = ctypes.c_int32(), ctypes.c_int32()
    if lib().oracle_geom_f32(d, ctypes.byref(L), ctypes.byref(V)) != 0:
        raise ValueError(f"no fp32 row geometry for d={d}")
    return L.value, V.value


def c_scores(corpus: np.ndarray, q: np.ndarray, *, order: str = "strict") -> np.ndarray:
    corpus = _f32(corpus)
    q = _f32(q).reshape(-1)
    out = np.empty(corpus.shape[0], dtype=np.float32)
    rc = lib().oracle_scores_order(_ptr(corpus, _f32p), corpus.shape[0], corpus.shape[1], _ptr(q, _f32p),
                                   ORDERS[order], _ptr(out, _f32p))
    if rc != 0:
        raise RuntimeError("oracle_scores_order failed")
    return out


def c_search(corpus: np.ndarray, q: np.ndarray, k: int, *, base: int = 0, f16: bool = False,
             normalize: bool = False, fast: bool = False, order: str = "strict"
             ) -> tuple[np.ndarray, np.ndarray]:
    """Exact inner-product top-k: (D f32 [nq,k], I i64 [nq,k]), best first, -1 padded.
    ``order`` picks the fp32 summation order (see ORDERS above; fp32 storage only)."""
    corpus = _f32(corpus)
    q = _f32(q)
    if q.ndim == 1:
        q = q[None, :]
    if corpus.ndim != 2:
        corpus = corpus.reshape(0, q.shape[1])
    if normalize:
        q = c_normalize_l2(q)
    if f16:
        corpus = c_round_f16(corpus)
        q = c_round_f16(q)
    n, d = corpus.shape
    nq = q.shape[0]
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    if order != "strict" and (f16 or fast):
        raise ValueError("kernel-order modes restate the fp32 kernels only (strict build)")
    rc = lib(fast).oracle_flat_ip_topk_order(_ptr(corpus, _f32p), n, d, _ptr(q, _f32p), nq, k, base,
                                             ORDERS[order], _ptr(D, _f32p), _ptr(I, _i64p))
    if rc != 0:
        raise RuntimeError("oracle_flat_ip_topk_order failed")
    return D, I


def c_merge(D_in: np.ndarray, I_in: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Merge [g, nq, k] per-shard results into [nq, k]."""
    D_in = _f32(D_in)
    I_in = np.ascontiguousarray(I_in, dtype=np.int64)
    g, nq, k = D_in.shape
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    rc = lib().oracle_merge_topk(_ptr(D_in, _f32p), _ptr(I_in, _i64p), g, nq, k, _ptr(D, _f32p),
                                 _ptr(I, _i64p))
    if rc != 0:
        raise RuntimeError("oracle_merge_topk failed")
    return D, I


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def set_num_threads(n: int) -> None:
    lib().oracle_set_num_threads(int(n))
    lib(True).oracle_set_num_threads(int(n))


# --------------------------------------------------------------------------- numpy twin
def np_normalize_l2(x: np.ndarray) -> np.ndarray:
    y = _f32(x).copy()
    if y.ndim == 1:
        y = y[None, :]
    nr = np.sqrt((y.astype(np.float64) ** 2).sum(axis=1))
    nz = nr > 0
    y[nz] = (y[nz] / nr[nz, None]).astype(np.float32)
    return y


def np_search(corpus: np.ndarray, q: np.ndarray, k: int, *, base: int = 0, f16: bool = False
              ) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """float64 twin. Returns (D f32 [nq,k], I i64 [nq,k], S f64 [nq,n] all scores)."""
    corpus = _f32(corpus)
    q = _f32(q)
    if q.ndim == 1:
        q = q[None, :]
    if f16:
        corpus = corpus.astype(np.float16).astype(np.float32)
        q = q.astype(np.float16).astype(np.float32)
    n = corpus.shape[0]
    nq = q.shape[0]
    S = q.astype(np.float64) @ corpus.astype(np.float64).T if n else np.zeros((nq, 0))
    D = np.full((nq, k), NEG_FLT_MAX, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    for i in range(nq):
        s32 = S[i].astype(np.float32)
        valid = np.nonzero(s32 > NEG_FLT_MAX)[0]  # drops NaN / -inf / -FLT_MAX
        order = valid[np.lexsort((valid, -s32[valid].astype(np.float64)))][:k]
        D[i, : order.size] = s32[order]
        I[i, : order.size] = order + base
    return D, I, S


# --------------------------------------------------------------------------- comparison
def compare_topk(D_a: np.ndarray, I_a: np.ndarray, D_ref: np.ndarray, I_ref: np.ndarray,
                 S_ref: np.ndarray | None = None, *, score_tol: float = 1e-5,
                 tie_eps: float = 2e-6, base: int = 0) -> dict:
    """Compare a result against a reference result.

    Scores must agree rank-by-rank within ``score_tol`` (BASELINE.json: 1e-5 on fp32 scores).
    Indices must be identical rank-by-rank, except where the reference's own scores for the two
    differing rows are within ``tie_eps`` of each other (a near-tie whose order legitimately
    depends on fp32 summation order; needs ``S_ref``, the reference's full score matrix).
    Returns counts; raises AssertionError on a real mismatch.
    """
    assert D_a.shape == D_ref.shape and I_a.shape == I_ref.shape
    pad = I_ref < 0
    assert np.array_equal(I_a < 0, pad), "padding (-1) positions differ"
    ok = ~pad
    ds = np.abs(D_a[ok].astype(np.float64) - D_ref[ok].astype(np.float64))
    max_ds = float(ds.max()) if ds.size else 0.0
    assert max_ds <= score_tol, f"score mismatch {max_ds} > {score_tol}"
    assert np.array_equal(D_a[pad], D_ref[pad]), "padding scores differ"
    diff = (I_a != I_ref) & ok
    n_diff = int(diff.sum())
    n_excused = 0
    if n_diff:
        assert S_ref is not None, f"{n_diff} index mismatches and no score matrix to excuse ties"
        for qi, j in zip(*np.nonzero(diff)):
            sa = S_ref[qi, I_a[qi, j] - base]
            sr = S_ref[qi, I_ref[qi, j] - base]
            assert abs(sa - sr) <= tie_eps, (
                f"query {qi} rank {j}: got row {I_a[qi, j]} (ref score {sa}) want "
                f"{I_ref[qi, j]} (ref score {sr})")
            n_excused += 1
    # recall@k. A row the reference does not list still counts when the reference's own score for
    # it is within tie_eps of the reference's k-th score: a near-tie AT the rank-k boundary, where
    # either row is a correct k-th result (the rank-by-rank check above has already verified that
    # this is the only way the two lists differ).
    recall = 1.0
    n_boundary = 0
    if ok.any():
        hits = 0
        for i in range(I_a.shape[0]):
            a = set(I_a[i][I_a[i] >= 0].tolist())
            r = set(I_ref[i][I_ref[i] >= 0].tolist())
            hits += len(a & r)
            if S_ref is not None and r and a - r:
                kth = min(float(S_ref[i, j - base]) for j in r)
                tied = sum(1 for j in a - r if float(S_ref[i, j - base]) >= kth - tie_eps)
                hits += tied
                n_boundary += tied
        recall = hits / int(ok.sum())
    return {"max_score_err": max_ds, "index_mismatches": n_diff, "near_ties_excused": n_excused,
            "boundary_ties": n_boundary, "recall": recall}


def compare_kernel_order(D_a: np.ndarray, I_a: np.ndarray, corpus: np.ndarray, q: np.ndarray, k: int, *,
                         base: int = 0, orders: tuple[str, ...] | None = None) -> dict:
    """ZERO-EXCUSE check for an fp32 index: every query's scores AND indices must be bit-identical to the
    oracle run in the library's own documented summation order - no tolerance, no near-tie excuse.

    ``q`` is the query as the kernels see it (already normalised if the call normalised). ``orders`` names
    the admissible orders per query; default: what the library documents for a call of this size - "scan"
    for nq <= 23, and for larger batches "fma" (the f32 MFMA pass) or, per query, "scan" (a query the
    batched path re-ran through its exact scan path, or a batch it handed to the scan path whole).
    Returns how many queries matched each order; raises AssertionError on the first query that matches
    none, naming the first differing rank."""
    q = _f32(q)
    if q.ndim == 1:
        q = q[None, :]
    nq = q.shape[0]
    if orders is None:
        orders = ("scan",) if nq <= SCAN_PATH_MAX_NQ_F32 else ("fma", "scan")
    refs = {o: c_search(corpus, q, k, base=base, order=o) for o in orders}
    matched = {o: 0 for o in orders}
    for i in range(nq):
        for o in orders:
            Dr, Ir = refs[o]
            if np.array_equal(I_a[i], Ir[i]) and np.array_equal(D_a[i], Dr[i]):
                matched[o] += 1
                break
        else:
            Dr, Ir = refs[orders[0]]
            bad = np.nonzero((I_a[i] != Ir[i]) | (D_a[i] != Dr[i]))[0]
            j = int(bad[0])
            raise AssertionError(
                f"query {i}: not bit-identical to the {'/'.join(orders)} order: first difference at rank {j}: "
                f"got (row {I_a[i, j]}, score {D_a[i, j]!r}) want (row {Ir[i, j]}, score {Dr[i, j]!r}); "
                f"{bad.size} of {k} ranks differ")
    return {"kernel_order_queries": matched, "kernel_order_mismatches": 0}
