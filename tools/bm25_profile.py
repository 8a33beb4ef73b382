"""Where a NameRetriever call spends its time (tokenise / GPU search / result dict)."""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np
from lean_explore_amd.bm25 import NameRetriever
from lean_explore_amd.search.tokenization import tokenize_raw, tokenize_spaced
W = ("continuous function compact set prime number group ring field ideal module measure "
     "integral limit sequence series norm metric topology open closed bounded linear map "
     "kernel image finite infinite sum product order lattice filter basis dimension").split()
NW = len(W)
n = 200_000
names = [f"Mathlib.{W[i % NW].capitalize()}.{W[(i * 7) % NW]}_{W[(i * 13) % NW]}_{i}" for i in range(n)]
ids = list(range(1000, 1000 + n))
lex = NameRetriever.from_names(ids, names)
q = " ".join(W[(3 * (j + 5) + j) % NW] for j in range(8))
def t(fn, R=200):
    fn(); t0 = time.perf_counter()
    for _ in range(R): fn()
    return (time.perf_counter() - t0) / R * 1e6
ts, tr = tokenize_spaced(q), tokenize_raw(q)
print("tokens", ts, tr)
print(f"tokenize both      {t(lambda: (tokenize_spaced(q), tokenize_raw(q))):8.1f} us")
print(f"retrieve spaced    {t(lambda: lex.spaced.retrieve(ts, 1000)):8.1f} us  postings={sum(int(lex.spaced.indptr[i+1]-lex.spaced.indptr[i]) for i in lex.spaced.token_ids(ts))}")
print(f"retrieve raw       {t(lambda: lex.raw.retrieve(tr, 1000)):8.1f} us")
print(f"whole call         {t(lambda: lex(q, 1000)):8.1f} us")
