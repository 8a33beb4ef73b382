import sys, time; sys.path.insert(0, __import__('os').path.abspath(__import__('os').path.join(__import__('os').path.dirname(__file__), '..', '..')))
import numpy as np
from lean_explore_amd.bm25 import BM25Index
from lean_explore_amd.search.tokenization import tokenize_spaced
from tests.test_bm25 import synth_names
names=synth_names(200_000,3); corpus=[list(dict.fromkeys(tokenize_spaced(n))) for n in names]
ix=BM25Index().index(corpus)
for q, k in ((["nat","add","comm"], 1000), (["nat","add","comm"], 50), (["measure", "theory"], 1000)):
    for rnd in range(2):
        for mode in (0, 1):
            ix.debug_option(0, mode)
            for _ in range(30): ix.retrieve(q,k)
            lat=[]
            for _ in range(300):
                t0=time.perf_counter(); ix.retrieve(q,k); lat.append(time.perf_counter()-t0)
            print(f"bm25 q={q} k={k} same_launch={mode}: p50 {np.median(lat)*1e6:.1f} us, mean {np.mean(lat)*1e6:.1f}; retries {ix.debug_counter(8)}, left fast path {ix.debug_counter(0)}", flush=True)
