import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
from lean_explore_amd.bm25 import BM25Index
from lean_explore_amd.search.tokenization import tokenize_spaced
from tests.test_bm25 import synth_names
names=synth_names(200_000,3); corpus=[list(dict.fromkeys(tokenize_spaced(n))) for n in names]
ix=BM25Index().index(corpus)
q=["nat","add","comm"]; ids=ix.token_ids(q)
nnz=sum(int(ix.indptr[t+1]-ix.indptr[t]) for t in ids)
for _ in range(20): ix.retrieve(q,1000)
t0=time.perf_counter(); K=300
for _ in range(K): ix.retrieve(q,1000)
dt=(time.perf_counter()-t0)/K
print(f"bm25 N=200k query={q} postings={nnz}: {dt*1e6:.1f} us/query (host API, sync) {1/dt:.0f} QPS; bytes={nnz*8+200000*12}")
print("selection left the fast path:", ix.debug_counter(0), "of", 320, "queries; general path:", ix.debug_counter(1))
ph=[ix.debug_counter(i)/100.0 for i in range(2,8)]
if any(ph): print("selection phases of a -DLS_FIN_TIMING build (us): loads issued + pivot", ph[0], "| survivors to LDS", ph[1], "| sample ranked + bucket counts + scan", ph[2],
                  "| scatter", ph[3], "| order", ph[4], "| output + done", ph[5])
