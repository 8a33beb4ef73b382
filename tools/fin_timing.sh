#!/bin/bash
# phase times of the standalone finalize kernel (build with -DLS_FIN_TIMING; 100 MHz ticks)
cd "$(dirname "$0")/.."
(cd lean-explore_amd/csrc && rm -f _build/ls_select.o _build/ls_scan.o _build/ls_bm25.o && make -s CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -DLS_FIN_TIMING" >/dev/null 2>&1)
python - <<'PY'
import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
for d,k in ((384,50),(1024,1000)):
    c=H.gauss(1234,200000,d); ix=FlatIPIndex.from_array(c)
    acc=np.zeros(5)
    for i in range(30):
        ix.search(H.gauss(100+i,1,d),k)
        if i>=10: acc+=np.array([ix.debug_counter(2+j) for j in range(5)])
    print(f"d={d} k={k}: load {acc[0]/200:.2f} us, radix passes {acc[1]/200:.2f}, compaction {acc[2]/200:.2f}, order {acc[3]/200:.2f}, output {acc[4]/200:.2f}")
    ix.close()
PY
(cd lean-explore_amd/csrc && rm -f _build/ls_select.o _build/ls_scan.o _build/ls_bm25.o && make -s >/dev/null 2>&1)
