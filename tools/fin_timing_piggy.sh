#!/bin/bash
# phase times of the piggy-backed finalize (256 threads, inside the next scan launch) on small shards
cd "$(dirname "$0")/.."
(cd lean-explore_amd/csrc && rm -f _build/ls_select.o _build/ls_scan.o _build/ls_bm25.o && make -s CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -DLS_FIN_TIMING" >/dev/null 2>&1)
python - <<'PY'
import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
d,k=384,50
for n in (10_000, 25_000, 200_000):
    ix=FlatIPIndex.from_array(H.gauss(1234,n,d))
    q=torch.from_numpy(H.gauss(5,1,d)).cuda()
    o=[(torch.empty((1,k),device="cuda"),torch.empty((1,k),dtype=torch.int64,device="cuda")) for _ in range(4)]
    acc=np.zeros(5); m=0
    for i in range(40):
        ix.search_device(q,k,*o[i&3],pipeline=True)
        if i>=10 and i%3==0:
            torch.cuda.synchronize(); acc+=np.array([ix.debug_counter(2+j) for j in range(5)]); m+=1
    ix.check()
    a=acc/m/100
    print(f"N={n}: load {a[0]:.2f} us, radix {a[1]:.2f}, compaction {a[2]:.2f}, order {a[3]:.2f}, output {a[4]:.2f}  total {a.sum():.2f}")
    ix.close()
PY
(cd lean-explore_amd/csrc && rm -f _build/ls_select.o _build/ls_scan.o _build/ls_bm25.o && make -s >/dev/null 2>&1)
