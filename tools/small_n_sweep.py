"""Step time of pipelined batch-1 searches on small shards (what an 8-GPU shard of config 2 sees)."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from lean_explore_amd.index import FlatIPIndex
d, k = 384, 50
rng = np.random.default_rng(0)
for n in (10_000, 25_000, 50_000, 100_000):
    c = rng.standard_normal((n, d), dtype=np.float32)
    ix = FlatIPIndex.from_array(c)
    q = torch.from_numpy(rng.standard_normal((1, d), dtype=np.float32)).cuda()
    outs = [(torch.empty((1, k), device="cuda"), torch.empty((1, k), dtype=torch.int64, device="cuda")) for _ in range(16)]
    for i in range(300): ix.search_device(q, k, *outs[i & 15], pipeline=True)
    ix.check(); torch.cuda.synchronize(); t = time.perf_counter()
    R = 10000
    for i in range(R): ix.search_device(q, k, *outs[i & 15], pipeline=True)
    ix.check(); torch.cuda.synchronize()
    print(f"N={n}: {(time.perf_counter()-t)/R*1e6:.2f} us/step")
