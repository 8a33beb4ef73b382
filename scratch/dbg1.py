import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
from oracle import oracle
for (n,d) in [(2000,384),(2000,768),(2000,64)]:
    c=H.gauss(1234,n,d); q=H.gauss(5678,1,d)
    ix=FlatIPIndex.from_array(c)
    D,I=ix.search(q,10)
    S=ix.debug_scores()
    Sr=(c.astype(np.float64)@q[0].astype(np.float64))
    print(n,d,"S maxerr",np.abs(S-Sr).max(), "bad rows", np.nonzero(np.abs(S-Sr)>1e-5)[0][:20])
    Dr,Ir=oracle.c_search(c,q,10)
    print(" I ",I[0]); print(" Ir",Ir[0]); print(" D",D[0][:5],Dr[0][:5], "slow", ix.debug_counter(0))
    ix.debug_option(1,1); D2,I2=ix.search(q,10); print(" slow I", I2[0])
