import sys, time; sys.path.insert(0, __import__('os').path.abspath(__import__('os').path.join(__import__('os').path.dirname(__file__), '..', '..')))
import numpy as np
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
for d, k in ((384, 1000), (384, 300), (1024, 1000)):
    ix = FlatIPIndex.from_array(H.gauss(1234, 200_000, d))
    for nq in (16, 17, 32):
        before = ix.debug_counter(25)
        calls = 60
        for c in range(calls):
            ix.search(H.gauss(1000 + c, nq, d), k)
        r = ix.debug_counter(25) - before
        print(f"retry d={d} k={k} nq={nq}: {r} second serves in {calls * nq} queries = {r / (calls * nq):.4f} per query", flush=True)
    ix.close()
