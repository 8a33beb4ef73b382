"""TEST INFRASTRUCTURE ONLY — numpy restatement of the lexical retriever of LeanExplore's local
backend: two BM25+ indices over declaration names, k = 1000 each, max-merged
(reference src/lean_explore/search/engine.py:175-223; built at
src/lean_explore/extract/index.py:238-266 with ``bm25s.BM25(method="bm25+")``).

The arithmetic lives in the third-party **bm25s** (reference pyproject.toml:39, constraint
">=0.2.0", no lockfile, not vendored, not installed here). Its published algorithm (Lù 2024,
"BM25S: orders of magnitude faster lexical search via eager sparse scoring") is restated:

  build   df[t] = #docs containing t;  idf[t] = ln((N + 1) / df[t])                 (bm25+ idf)
          tfc(tf, |d|) = (k1 + 1) tf / (k1 (1 - b + b |d| / avgdl) + tf) + delta    (bm25+ tf part)
          nonocc[t] = idf[t] * delta          (score of a document that lacks t)
          CSC matrix, one column per token: rows = documents containing t,
          data = idf[t] * tfc(tf, |d|) - nonocc[t]                                   (float32)
  query   scores = 0 (float32); for each query token in order: scores[rows] += data;
          scores += sum(nonocc[query tokens]);  top-k by score
  defaults k1 = 1.5, b = 0.75, delta = 0.5.

PARITY UNPINNED: the reference's tests hold no BM25 numbers (tokenisers only,
tests/search/engine_test.py:42-66) and bm25s cannot be imported; float rounding points (float64
arithmetic, one cast to float32 per stored value) and the tie order (score desc, doc asc) are
fixed by definition here. `brute_force_scores` recomputes BM25+ from the textbook formula as an
independent cross-check of the eager-sparse construction.
"""

from __future__ import annotations

import math

import numpy as np

K1, B, DELTA = 1.5, 0.75, 0.5


def build(corpus_tokens: list[list[str]], k1: float = K1, b: float = B, delta: float = DELTA) -> dict:
    """Eager BM25+ index. corpus_tokens[d] = token list of document d (the reference passes
    ``list(set(tokens))``, extract/index.py:255-256, so tf is 1 there; general tf is handled)."""
    vocab: dict[str, int] = {}
    for doc in corpus_tokens:
        for tok in doc:
            if tok not in vocab:
                vocab[tok] = len(vocab)
    n_docs, n_vocab = len(corpus_tokens), len(vocab)
    doc_len = np.array([len(d) for d in corpus_tokens], dtype=np.float64)
    avgdl = float(doc_len.mean()) if n_docs else 0.0
    rows, cols, tfs = [], [], []
    df = np.zeros(n_vocab, dtype=np.int64)
    for d, doc in enumerate(corpus_tokens):
        counts: dict[int, int] = {}
        for tok in doc:
            counts[vocab[tok]] = counts.get(vocab[tok], 0) + 1
        for t, c in counts.items():
            rows.append(d)
            cols.append(t)
            tfs.append(c)
            df[t] += 1
    idf = np.array([math.log((n_docs + 1) / x) if x else 0.0 for x in df], dtype=np.float64)
    nonocc = (idf * delta).astype(np.float32)
    rows = np.array(rows, dtype=np.int64)
    cols = np.array(cols, dtype=np.int64)
    tfs = np.array(tfs, dtype=np.float64)
    tfc = (k1 + 1.0) * tfs / (k1 * (1.0 - b + b * doc_len[rows] / avgdl) + tfs) + delta if rows.size \
        else np.zeros(0)
    vals = (idf[cols] * tfc - nonocc[cols].astype(np.float64)).astype(np.float32)
    order = np.lexsort((rows, cols))  # CSC: by column (token), then row (document)
    indptr = np.zeros(n_vocab + 1, dtype=np.int64)
    np.add.at(indptr, cols + 1, 1)
    indptr = np.cumsum(indptr)
    return {"vocab": vocab, "indptr": indptr.astype(np.int32), "indices": rows[order].astype(np.int32),
            "data": vals[order], "nonocc": nonocc, "n_docs": n_docs,
            "params": {"k1": k1, "b": b, "delta": delta, "method": "bm25+", "num_docs": n_docs,
                       "avgdl": avgdl}}


def token_ids(index: dict, query_tokens: list[str]) -> np.ndarray:
    """Query tokens -> ids; tokens outside the vocabulary are dropped (duplicates kept)."""
    v = index["vocab"]
    return np.array([v[t] for t in query_tokens if t in v], dtype=np.int32)


def scores(index: dict, ids: np.ndarray) -> np.ndarray:
    s = np.zeros(index["n_docs"], dtype=np.float32)
    for t in ids:
        a, b = index["indptr"][t], index["indptr"][t + 1]
        s[index["indices"][a:b]] += index["data"][a:b]   # one column: no repeated document
    shift = np.float32(0.0)
    for t in ids:
        shift = np.float32(shift + index["nonocc"][t])
    return s + shift


def retrieve(index: dict, query_tokens: list[str], k: int) -> tuple[np.ndarray, np.ndarray]:
    """(docs int64 [k], scores f32 [k]) best first under (score desc, doc asc); -1 / -FLT_MAX
    padding when k exceeds the number of documents."""
    s = scores(index, token_ids(index, query_tokens))
    order = np.lexsort((np.arange(s.size), -s.astype(np.float64)))[:k]
    docs = np.full(k, -1, dtype=np.int64)
    out = np.full(k, np.float32(-3.4028234663852886e38), dtype=np.float32)
    docs[: order.size] = order
    out[: order.size] = s[order]
    return docs, out


def brute_force_scores(corpus_tokens: list[list[str]], query_tokens: list[str], k1: float = K1,
                       b: float = B, delta: float = DELTA) -> np.ndarray:
    """Textbook BM25+ in float64: sum over query tokens of idf * (tf part + delta)."""
    n = len(corpus_tokens)
    avgdl = sum(len(d) for d in corpus_tokens) / n
    out = np.zeros(n, dtype=np.float64)
    present = set(t for d in corpus_tokens for t in d)
    for tok in query_tokens:
        if tok not in present:
            continue
        df = sum(1 for d in corpus_tokens if tok in d)
        idf = math.log((n + 1) / df)
        for i, d in enumerate(corpus_tokens):
            tf = d.count(tok)
            out[i] += idf * ((k1 + 1) * tf / (k1 * (1 - b + b * len(d) / avgdl) + tf) + delta)
    return out
