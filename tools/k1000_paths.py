import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
c=H.gauss(1234,200000,1024); 
ix=FlatIPIndex.from_array(c)
for i in range(50): ix.search(H.gauss(100+i,1,1024),1000)
print("left fast path:", ix.debug_counter(0), "general:", ix.debug_counter(1))
