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
