"""BM25+ name retrieval on the MI355X — the lexical half of the hybrid search
(reference src/lean_explore/search/engine.py:175-223).

`BM25Index` is the counterpart of a `bm25s.BM25(method="bm25+")` object: `index(corpus_tokens)`
builds bm25s's eager-sparse score matrix on the host (numpy, one-time, like the reference's
offline builder at src/lean_explore/extract/index.py:238-266), `save` / `load` use bm25s's
on-disk layout (reference src/lean_explore/cli/data_commands.py:42-59), and `retrieve` runs the
query on the GPU through the C ABI (`ls_bm25_*`, include/leansearch.h). `NameRetriever` is the
callable `SearchEngine(lexical_retriever=...)` expects: two indices (spaced / raw tokens),
k = 1000 each, max-merged per declaration id (engine.py:192-223).

bm25s is not installed here, so its formulas are restated from the published algorithm
(float64 arithmetic, one cast to float32 per stored score; PARITY UNPINNED — see
oracle/bm25_ref.py).
"""

from __future__ import annotations

import ctypes
import json
import math
from pathlib import Path

import numpy as np

from . import native
from .search.tokenization import tokenize_raw, tokenize_spaced

_FILES = {"data": "data.csc.index.npy", "indices": "indices.csc.index.npy",
          "indptr": "indptr.csc.index.npy", "nonocc": "nonoccurrence_array.index.npy",
          "params": "params.index.json", "vocab": "vocab.index.json"}


class BM25Index:
    def __init__(self, k1: float = 1.5, b: float = 0.75, delta: float = 0.5, method: str = "bm25+",
                 device: int = 0):
        if method != "bm25+":
            raise ValueError("only method='bm25+' (what the reference builds) is implemented")
        self.k1, self.b, self.delta, self.method, self.device = k1, b, delta, method, device
        self.vocab: dict[str, int] = {}
        self.indptr = np.zeros(1, np.int64)
        self.indices = np.zeros(0, np.int32)
        self.data = np.zeros(0, np.float32)
        self.nonoccurrence = np.zeros(0, np.float32)
        self.num_docs = 0
        self._handle: ctypes.c_void_p | None = None

    # ------------------------------------------------------------------ build (host, one-time)
    def index(self, corpus_tokens: list[list[str]]) -> "BM25Index":
        """bm25s `index(corpus_tokens)`: df, idf = ln((N+1)/df), CSC of idf*tfc - nonoccurrence."""
        self._drop()
        vocab: dict[str, int] = {}
        rows: list[int] = []
        cols: list[int] = []
        tfs: list[int] = []
        for d, doc in enumerate(corpus_tokens):
            counts: dict[int, int] = {}
            for tok in doc:
                t = vocab.setdefault(tok, len(vocab))
                counts[t] = counts.get(t, 0) + 1
            rows.extend([d] * len(counts))
            cols.extend(counts.keys())
            tfs.extend(counts.values())
        n_docs, n_vocab = len(corpus_tokens), len(vocab)
        r = np.asarray(rows, dtype=np.int64)
        c = np.asarray(cols, dtype=np.int64)
        tf = np.asarray(tfs, dtype=np.float64)
        doc_len = np.fromiter((len(d) for d in corpus_tokens), dtype=np.float64, count=n_docs)
        avgdl = float(doc_len.mean()) if n_docs else 0.0
        df = np.bincount(c, minlength=n_vocab).astype(np.float64)
        idf = np.where(df > 0, np.log((n_docs + 1) / np.maximum(df, 1.0)), 0.0)
        nonocc = (idf * self.delta).astype(np.float32)
        if r.size:
            tfc = (self.k1 + 1.0) * tf / (self.k1 * (1.0 - self.b + self.b * doc_len[r] / avgdl) + tf) \
                + self.delta
            vals = (idf[c] * tfc - nonocc[c].astype(np.float64)).astype(np.float32)
        else:
            vals = np.zeros(0, np.float32)
        order = np.lexsort((r, c))
        self.vocab = vocab
        self.indptr = np.concatenate([[0], np.cumsum(np.bincount(c, minlength=n_vocab))]).astype(np.int64)
        self.indices = r[order].astype(np.int32)
        self.data = vals[order]
        self.nonoccurrence = nonocc
        self.num_docs = n_docs
        self._avgdl = avgdl
        return self

    # ------------------------------------------------------------------ bm25s on-disk layout
    def save(self, directory: str | Path) -> None:
        d = Path(directory)
        d.mkdir(parents=True, exist_ok=True)
        np.save(d / _FILES["data"], self.data)
        np.save(d / _FILES["indices"], self.indices)
        np.save(d / _FILES["indptr"], self.indptr.astype(np.int32) if self.indptr[-1] < 2**31
                else self.indptr)
        np.save(d / _FILES["nonocc"], self.nonoccurrence)
        (d / _FILES["params"]).write_text(json.dumps({
            "k1": self.k1, "b": self.b, "delta": self.delta, "method": self.method,
            "idf_method": self.method, "dtype": "float32", "int_dtype": "int32",
            "num_docs": self.num_docs, "version": "lean_explore_amd"}))
        (d / _FILES["vocab"]).write_text(json.dumps(self.vocab))

    @classmethod
    def load(cls, directory: str | Path, device: int = 0) -> "BM25Index":
        d = Path(directory)
        params = json.loads((d / _FILES["params"]).read_text())
        ix = cls(k1=params.get("k1", 1.5), b=params.get("b", 0.75), delta=params.get("delta", 0.5),
                 method=params.get("method", "bm25+"), device=device)
        ix.data = np.ascontiguousarray(np.load(d / _FILES["data"]), dtype=np.float32)
        ix.indices = np.ascontiguousarray(np.load(d / _FILES["indices"]), dtype=np.int32)
        ix.indptr = np.ascontiguousarray(np.load(d / _FILES["indptr"]), dtype=np.int64)
        ix.nonoccurrence = np.ascontiguousarray(np.load(d / _FILES["nonocc"]), dtype=np.float32)
        ix.vocab = {str(k): int(v) for k, v in json.loads((d / _FILES["vocab"]).read_text()).items()}
        ix.num_docs = int(params["num_docs"])
        if ix.nonoccurrence.size != len(ix.vocab) and ix.nonoccurrence.size == 0:
            ix.nonoccurrence = np.zeros(ix.indptr.size - 1, np.float32)
        return ix

    # ------------------------------------------------------------------ GPU retrieval
    def _ensure(self) -> ctypes.c_void_p:
        if self._handle is None:
            lib = native.load()
            h = ctypes.c_void_p()
            n_vocab = self.indptr.size - 1
            native.check(lib.ls_bm25_create(
                ctypes.byref(h), self.indptr.ctypes.data,
                self.indices.ctypes.data if self.indices.size else None,
                self.data.ctypes.data if self.data.size else None,
                self.nonoccurrence.ctypes.data if n_vocab else None,
                self.num_docs, n_vocab, self.device))
            self._handle = h
        return self._handle

    def _drop(self) -> None:
        if self._handle is not None:
            native.load().ls_bm25_destroy(self._handle)
            self._handle = None

    def token_ids(self, query_tokens: list[str]) -> np.ndarray:
        """Tokens outside the vocabulary are dropped; duplicates are kept (bm25s semantics)."""
        v = self.vocab
        return np.fromiter((v[t] for t in query_tokens if t in v), dtype=np.int32)

    def retrieve(self, query_tokens: list[str], k: int) -> tuple[np.ndarray, np.ndarray]:
        """(docs int64 [k], scores float32 [k]), best first; (-1, -FLT_MAX) padded."""
        ids = np.ascontiguousarray(self.token_ids(query_tokens))
        docs = np.empty(k, dtype=np.int64)
        scores = np.empty(k, dtype=np.float32)
        native.check(native.load().ls_bm25_search(
            self._ensure(), native.addr(ids), ids.size, int(k), native.addr(scores),
            native.addr(docs)))
        return docs, scores

    def get_scores_host(self, query_tokens: list[str]) -> np.ndarray:
        """bm25s `get_scores`: the score of EVERY document, float32, accumulated column by column in
        query-token order on the host — the same additions in the same order as the GPU kernel
        (and as bm25s). For the throw-away index over the <= 50 rerank candidates (reference
        search/engine.py:418-448, computed with bm25s on the CPU there as well): creating a GPU handle
        for 50 documents cost 1.6 ms per query, this costs ~0.1 ms. The name indices (200 k documents,
        top-1000 selection) stay on the GPU."""
        s = np.zeros(self.num_docs, dtype=np.float32)
        shift = np.float32(0.0)
        for t in self.token_ids(query_tokens).tolist():
            a, b = int(self.indptr[t]), int(self.indptr[t + 1])
            s[self.indices[a:b]] += self.data[a:b]   # a column lists a document at most once
            shift = np.float32(shift + self.nonoccurrence[t])
        return s + shift

    def debug_counter(self, which: int) -> int:
        """0: searches whose selection left the fast path; 1: those that took the general select."""
        return int(native.load().ls_bm25_debug_counter(self._ensure(), which))

    def close(self) -> None:
        self._drop()

    def __del__(self):
        try:
            self._drop()
        except Exception:
            pass


class NameRetriever:
    """`_retrieve_bm25_candidates` of the reference (engine.py:192-223): spaced + raw token
    indices over declaration names, max-merged into {declaration id: score}."""

    def __init__(self, spaced: BM25Index, raw: BM25Index, declaration_ids: list[int]):
        self.spaced, self.raw, self.ids = spaced, raw, list(declaration_ids)

    @classmethod
    def from_names(cls, declaration_ids: list[int], names: list[str], device: int = 0
                   ) -> "NameRetriever":
        """Build both indices like the reference's builder (extract/index.py:255-263): each
        name's UNIQUE tokens."""
        spaced = BM25Index(device=device).index([list(dict.fromkeys(tokenize_spaced(n or "")))
                                                 for n in names])
        raw = BM25Index(device=device).index([list(dict.fromkeys(tokenize_raw(n or "")))
                                              for n in names])
        return cls(spaced, raw, declaration_ids)

    @classmethod
    def load(cls, base_path: str | Path, device: int = 0) -> "NameRetriever":
        """The reference's layout: bm25_name_spaced/, bm25_name_raw/, bm25_ids_map.json."""
        base = Path(base_path)
        ids = json.loads((base / "bm25_ids_map.json").read_text())
        return cls(BM25Index.load(base / "bm25_name_spaced", device),
                   BM25Index.load(base / "bm25_name_raw", device), ids)

    def save(self, base_path: str | Path) -> None:
        base = Path(base_path)
        self.spaced.save(base / "bm25_name_spaced")
        self.raw.save(base / "bm25_name_raw")
        (base / "bm25_ids_map.json").write_text(json.dumps(self.ids))

    def __call__(self, query: str, bm25_k: int) -> dict[int, float]:
        out: dict[int, float] = {}
        for index, tokens in ((self.spaced, tokenize_spaced(query)), (self.raw, tokenize_raw(query))):
            docs, scores = index.retrieve(tokens, min(bm25_k, max(1, index.num_docs)))
            ids, get = self.ids, out.get
            # plain Python numbers: iterating numpy scalars costs ~1 ms per 2000 results
            for doc, score in zip(docs.tolist(), scores.tolist()):
                if doc < 0:
                    continue
                decl_id = ids[doc]
                prev = get(decl_id, 0.0)
                out[decl_id] = score if score > prev else prev
        return out
