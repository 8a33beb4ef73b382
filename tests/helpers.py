"""Shared input builders for the parity tests (same recipes as tests/golden/make_golden.py
and BASELINE.md §3: seeded standard-normal rows, L2-normalised)."""

from __future__ import annotations

import hashlib
import json
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def gauss(seed: int, n: int, d: int, normalize: bool = True) -> np.ndarray:
    x = np.random.default_rng(seed).standard_normal((n, d), dtype=np.float32)
    if normalize and n:
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


def int_corpus(seed: int, n: int, d: int, lo: int = -3, hi: int = 4) -> np.ndarray:
    """Small-integer rows: every dot product is exact in fp32 (and fp16 storage), whatever the
    summation order, so scores AND indices must match the oracle bit for bit; ties abound."""
    return np.random.default_rng(seed).integers(lo, hi, size=(n, d)).astype(np.float32)


def load_golden():
    meta = json.loads((GOLDEN / "dense_golden.json").read_text())
    arrays = np.load(GOLDEN / "dense_golden.npz")
    return meta, arrays


def kat_inputs(seed: int):
    """reference tests/extract/index_test.py:186-205 (row 0 = e0, query = e0), fixed seed."""
    rng = np.random.default_rng(seed)
    emb = rng.random((300, 768), dtype=np.float32)
    emb[0] = 0.0
    emb[0, 0] = 1.0
    q = np.zeros((1, 768), np.float32)
    q[0, 0] = 1.0
    return emb, q


def write_ivf_flat(path, corpus: np.ndarray, assign: np.ndarray, nlist: int) -> None:
    """Write an ``IwFl`` file with the given row -> list assignment: exercises
    lean_explore_amd.faiss_compat.read_index on the container layout the reference ships
    (reference extract/index.py:103-104,173). Not a trained IVF; test infrastructure only."""
    import struct

    from lean_explore_amd import faiss_compat as fc

    corpus = np.ascontiguousarray(corpus, dtype="<f4")
    n, d = corpus.shape
    with open(path, "wb") as f:
        f.write(struct.pack("<I", fc._fourcc("IwFl")))
        fc._write_header(f, d, n, fc.METRIC_INNER_PRODUCT)
        f.write(struct.pack("<QQ", nlist, 1))
        f.write(struct.pack("<I", fc._fourcc("IxFI")))           # quantizer: nlist zero centroids
        fc._write_header(f, d, nlist, fc.METRIC_INNER_PRODUCT)
        f.write(struct.pack("<Q", nlist * d))
        f.write(np.zeros(nlist * d, "<f4").tobytes())
        f.write(struct.pack("<b", 0))                         # no direct map
        f.write(struct.pack("<Q", 0))
        f.write(struct.pack("<I", fc._fourcc("ilar")))
        f.write(struct.pack("<QQ", nlist, 4 * d))
        f.write(struct.pack("<I", fc._fourcc("full")))
        sizes = np.bincount(assign, minlength=nlist).astype("<u8")
        f.write(struct.pack("<Q", nlist))
        f.write(sizes.tobytes())
        for li in range(nlist):
            ids = np.nonzero(assign == li)[0].astype("<i8")
            if ids.size:
                f.write(corpus[ids].tobytes())
                f.write(ids.tobytes())
