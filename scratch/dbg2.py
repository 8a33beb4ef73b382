import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
n,d=64,64
c=np.arange(n*d,dtype=np.float32).reshape(n,d)
ix=FlatIPIndex.from_array(c)
for j in [0,1,3,4,5,8,16,63]:
    q=np.zeros((1,d),np.float32); q[0,j]=1
    ix.search(q,1)
    S=ix.debug_scores()
    print("j",j,"S[:6]",S[:6], "expected", c[:6,j])
q=np.ones((1,d),np.float32)
ix.search(q,1); S=ix.debug_scores(); print("ones", S[:4], c[:4].sum(1))
