"""Large query batch on small fp16 shards (what each rank of a many-GPU config-3 run sees)."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from lean_explore_amd.index import FlatIPIndex
d, k, nq = 384, 100, 1024
rng = np.random.default_rng(0)
q = torch.from_numpy(rng.standard_normal((nq, d), dtype=np.float32)).cuda()
for n in (8192, 12_500, 25_000, 50_000, 100_000):
    ix = FlatIPIndex.from_array(rng.standard_normal((n, d), dtype=np.float32), dtype="f16")
    for allow in (1, 0):
        ix.debug_option(4, allow)
        for _ in range(3): ix.search_device(q, k, asynchronous=True)
        ix.check(); torch.cuda.synchronize(); t = time.perf_counter(); R = 20
        for _ in range(R): ix.search_device(q, k, asynchronous=True)
        ix.check(); torch.cuda.synchronize()
        print(f"N={n} {'batched' if allow else 'scan   '}: {(time.perf_counter()-t)/R*1e6:.1f} us/batch  repaired={ix.debug_counter(8)}")
