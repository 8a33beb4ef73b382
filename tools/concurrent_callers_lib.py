"""run(): T threads making synchronous single-query ls_search calls on one handle (shared by the callers tools)."""
import threading, time
import numpy as np

def run(ix, q, k, T, secs=1.0):
    counts = [0] * T; lats = [[] for _ in range(T)]; stop = time.perf_counter() + secs
    def w(t):
        qq = q[t:t + 1]
        while time.perf_counter() < stop:
            t0 = time.perf_counter(); ix.search(qq, k, normalize=True); lats[t].append(time.perf_counter() - t0); counts[t] += 1
    th = [threading.Thread(target=w, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t0
    allat = np.concatenate([np.asarray(l) for l in lats])
    return sum(counts) / dt, np.median(allat) * 1e6

