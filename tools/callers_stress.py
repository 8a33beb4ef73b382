"""Exactness under concurrency: 12 threads, mixed k / normalize / 1-3 queries per call, callers that come and go; every row
must equal the lone call's (array_equal).  gpurun -- 'python tools/callers_stress.py'"""
import sys, threading, time, random; sys.path.insert(0, '/root/repo')
import numpy as np
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
for n, d in ((120_000, 384), (150_000, 1024)):
    c = H.gauss(1, n, d); pool = H.gauss(2, 24, d, normalize=False)
    ix = FlatIPIndex.from_array(c)
    ks = (10, 50, 100, 1000)
    want = {(i, k, nz): ix.search(pool[i:i + 1], k, normalize=nz) for i in range(24) for k in ks for nz in (False, True)}
    bad, cnt = [], [0]
    stop = time.perf_counter() + 6.0
    def w(t):
        r = random.Random(t)
        while time.perf_counter() < stop:
            i, k, nz = r.randrange(24), r.choice(ks), r.random() < 0.5
            m = r.choice((1, 1, 1, 2, 3))
            idx = [(i + j) % 24 for j in range(m)]
            D, I = ix.search(pool[idx] if m > 1 else pool[i:i + 1], k, normalize=nz)
            for j, ii in enumerate(idx):
                if not (np.array_equal(D[j], want[(ii, k, nz)][0][0]) and np.array_equal(I[j], want[(ii, k, nz)][1][0])):
                    bad.append((t, ii, k, nz, m))
            cnt[0] += 1
            if r.random() < 0.02: time.sleep(r.random() * 0.002)  # callers come and go
    th = [threading.Thread(target=w, args=(t,)) for t in range(12)]
    [x.start() for x in th]; [x.join() for x in th]
    print(f"d={d}: {cnt[0]} calls from 12 threads (mixed k, normalize, 1-3 queries per call), mismatches: {len(bad)} {bad[:3]}; combined batches {ix.debug_counter(16)}, requests in them {ix.debug_counter(17)}, served again {ix.debug_counter(25)}", flush=True)
    ix.close()
