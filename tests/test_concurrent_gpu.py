"""Concurrent synchronous callers on one handle (an MCP server with several clients; the reference's
call is one blocking index.search per query, search/engine.py:250): requests that arrive while a
search is running are served together, up to 16 queries per corpus pass, and every caller gets
exactly the rows and scores its own separate call would have produced."""

import threading

import numpy as np
import pytest

from lean_explore_amd.index import FlatIPIndex
from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices", [None, [0, 0, 0]])
def test_concurrent_callers_are_combined_and_exact(devices):
    n, d, T, per = 120_000, 256, 8, 30
    corpus = H.gauss(81, n, d)
    pool = H.gauss(82, 64, d, normalize=False)
    ix = FlatIPIndex.from_array(corpus, devices=devices) if devices else FlatIPIndex.from_array(corpus)
    try:
        ix.search(pool[:1], 10)  # build + warm
        # what each (query, k, normalize) must return: separate calls, combining off
        ix.debug_option(10, 0)
        want = {}
        for qi in range(64):
            for k, norm in ((50, True), (20, False)):
                want[(qi, k, norm)] = ix.search(pool[qi:qi + 1], k, normalize=norm)
        ix.debug_option(10, 1)
        errors = []

        def worker(t):
            rng = np.random.default_rng(t)
            for j in range(per):
                qi = int(rng.integers(64))
                k, norm = ((50, True), (20, False))[int(rng.integers(4) == 0)]
                two = j % 7 == 3  # some callers bring two queries
                q = pool[[qi, (qi + 1) % 64]] if two else pool[qi:qi + 1]
                D, I = ix.search(q, k, normalize=norm)
                for r, qq in enumerate([qi, (qi + 1) % 64][: q.shape[0]]):
                    Dw, Iw = want[(qq, k, norm)]
                    if not (np.array_equal(D[r], Dw[0]) and np.array_equal(I[r], Iw[0])):
                        errors.append((t, j, qq, k, norm))

        threads = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        assert not errors, errors[:5]
        assert ix.debug_counter(16) >= 1, "8 threads hammering one handle never met in a batch"
        assert ix.debug_counter(17) >= 2 * ix.debug_counter(16)
        # and the separate-call results themselves are the oracle's
        Dr, Ir = oracle.c_search(corpus, pool[:4], 50, normalize=True)
        _, _, S = oracle.np_search(corpus, oracle.c_normalize_l2(pool[:4]), 50)
        D = np.concatenate([want[(qi, 50, True)][0] for qi in range(4)])
        I = np.concatenate([want[(qi, 50, True)][1] for qi in range(4)])
        assert oracle.compare_topk(D, I, Dr, Ir, S)["recall"] == 1.0
    finally:
        ix.close()


def test_errors_reach_the_caller_that_made_them():
    from lean_explore_amd import native

    ix = FlatIPIndex.from_array(H.gauss(83, 5000, 64))
    try:
        out = []

        def bad():
            try:
                ix.search(H.gauss(84, 1, 64), 5000)  # min(k, ntotal) > LS_MAX_K
            except native.LeanSearchError as e:
                out.append(e.code)

        def good():
            D, I = ix.search(H.gauss(85, 1, 64), 5)
            out.append(int(I.shape[1]))

        ths = [threading.Thread(target=f) for f in (bad, good, bad, good)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert sorted(out) == sorted([native.LS_ERR_K_TOO_LARGE] * 2 + [5, 5])
    finally:
        ix.close()


def test_two_alternating_callers_overlap_and_stay_exact():
    """Round 5: a synchronous call runs in two phases on one of two host slots; with two callers taking turns
    the second call's launch is queued while the first still polls for its answer (debug counter 24 counts
    such calls). Every answer must be bit-identical to the caller's own separate call (zero-excuse oracle
    included), also while a third thread adds rows / checks / sets the base in between - those entry
    points wait for the host calls in flight (ls_quiesce) - and with the overlap switched off (option 17)."""
    n, d = 100_000, 384
    corpus = H.gauss(91, n, d)
    pool = H.gauss(92, 32, d)
    ix = FlatIPIndex.from_array(corpus)
    try:
        want = [ix.search(pool[i:i + 1], 50) for i in range(32)]
        oracle.compare_kernel_order(np.concatenate([w[0] for w in want[:4]]), np.concatenate([w[1] for w in want[:4]]),
                                    corpus, pool[:4], 50, orders=("scan",))
        # (gather: 0 = the two-deep overlap decides, round 5's first form; 2 = the shipped default, callers are
        # gathered into shared passes and nothing goes early)
        for overlap, gather in ((1, 0), (0, 0), (1, 2)):
            ix.debug_option(17, overlap)
            ix.debug_option(20, gather)
            before = ix.debug_counter(24)
            errors, stop = [], threading.Event()

            def worker(t):
                for j in range(300):
                    i = (7 * j + 13 * t) % 32
                    D, I = ix.search(pool[i:i + 1], 50)
                    if not (np.array_equal(D, want[i][0]) and np.array_equal(I, want[i][1])):
                        errors.append((t, j, i))

            def meddler():
                from lean_explore_amd import native

                lib = native.load()
                while not stop.is_set():
                    ix.check()
                    native.check(lib.ls_set_base(ix._handle, 0))
                    stop.wait(0.0005)  # (both take every host slot: leave the callers room to overlap)

            ths = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
            m = threading.Thread(target=meddler)
            m.start()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            stop.set()
            m.join()
            assert not errors, (overlap, gather, errors[:5])
            if overlap and not gather:
                assert ix.debug_counter(24) > before, "two alternating callers never overlapped"
            elif not overlap:
                assert ix.debug_counter(24) == before
    finally:
        ix.close()


def test_callers_of_long_passes_are_gathered_and_stay_exact():
    """d = 1024 (the reference's shape): a combined call takes > 110 us, so concurrent callers are gathered into
    one pass (debug option 20) instead of running two passes at once. Rows are those of lone calls, bit for bit."""
    import threading

    c = H.gauss(71, 200_000, 1024)
    q = H.gauss(72, 8, 1024)
    ix = FlatIPIndex.from_array(c)
    want = [ix.search(q[j:j + 1], 100, normalize=True) for j in range(8)]
    bad = []

    def w(t):
        for _ in range(60):
            D, I = ix.search(q[t:t + 1], 100, normalize=True)
            if not (np.array_equal(D, want[t][0]) and np.array_equal(I, want[t][1])):
                bad.append(t)

    b0, r0 = ix.debug_counter(16), ix.debug_counter(17)
    th = [threading.Thread(target=w, args=(t,)) for t in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not bad
    batches, reqs = ix.debug_counter(16) - b0, ix.debug_counter(17) - r0
    assert reqs >= 180 and reqs / batches > 2.5, (batches, reqs)  # (combined batches only) mostly passes of 3-4 callers
    ix.close()


@pytest.mark.parametrize("T", [28, 44])
def test_more_than_sixteen_callers_share_wide_passes_and_stay_exact(T):
    """Round 6: one ls_mq pass carries up to 32 queries (two MFMA B blocks), so 28 concurrent single-query callers -
    some bringing three queries - are served in passes of 17..32; with 44, a queue that alone fills a pass is
    launched behind the call in flight (two host slots). Every caller must get exactly what its own
    separate call returns, bit for bit, whatever company it rode in (reference: index.search from several MCP
    clients, search/engine.py:250, mcp/server.py:147-151)."""
    n, d, per = 100_000, 384, 25
    corpus = H.gauss(91, n, d)
    pool = H.gauss(92, 96, d, normalize=False)
    ix = FlatIPIndex.from_array(corpus)
    try:
        ix.search(pool[:1], 10)
        ix.debug_option(10, 0)  # separate calls
        want = {(qi, k): ix.search(pool[qi:qi + 1], k, normalize=True) for qi in range(96) for k in (50, 300)}
        ix.debug_option(10, 1)
        before_mq, before_launches = ix.debug_counter(23), ix.debug_counter(11)
        errors, sizes = [], []

        def worker(t):
            rng = np.random.default_rng(1000 + t)
            for j in range(per):
                qi = int(rng.integers(94))
                k = 300 if rng.integers(8) == 0 else 50
                m = 3 if j % 9 == 4 else 1
                D, I = ix.search(pool[qi:qi + m], k, normalize=True)
                for r in range(m):
                    Dw, Iw = want[(qi + r, k)]
                    if not (np.array_equal(D[r], Dw[0]) and np.array_equal(I[r], Iw[0])):
                        errors.append((t, j, qi + r, k))

        threads = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        assert not errors, errors[:5]
        batches, requests = ix.debug_counter(16), ix.debug_counter(17)
        assert batches >= 1 and requests > 2 * batches, (batches, requests)
        print(T, "callers:", requests, "requests in", batches, "combined batches;", ix.debug_counter(23) - before_mq, "ls_mq launches of",
              ix.debug_counter(11) - before_launches)
    finally:
        ix.close()


_SLEEPERS = r"""
import sys, threading
import numpy as np
sys.path.insert(0, {root!r})
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
n, d, T, per = 100_000, 384, 24, 30
ix = FlatIPIndex.from_array(H.gauss(93, n, d))
pool = H.gauss(94, 64, d, normalize=False)
ix.search(pool[:1], 10)
ix.debug_option(10, 0)
want = {{qi: ix.search(pool[qi:qi + 1], 50, normalize=True) for qi in range(64)}}
ix.debug_option(10, 1)
bad = []
def worker(t):
    rng = np.random.default_rng(t)
    for j in range(per):
        qi = int(rng.integers(64))
        D, I = ix.search(pool[qi:qi + 1], 50, normalize=True)
        if not (np.array_equal(D, want[qi][0]) and np.array_equal(I, want[qi][1])):
            bad.append((t, j, qi))
th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
[x.start() for x in th]; [x.join() for x in th]
print("RESULT", len(bad), ix.debug_counter(16), ix.debug_counter(17), ix.debug_counter(33))
ix.close()
"""


def test_sleeping_waiters_are_woken_by_their_batch_and_stay_exact():
    """Round 6: only as many waiters poll as the process has CPUs for (affinity mask, cgroup quota; LS_SPIN_CPUS
    overrides), the others sleep on their OWN request and are woken by the thread that finished their batch. With
    LS_SPIN_CPUS=2 ONE waiter may poll (max(1, cpus - 3)): 24 callers, nearly every one asleep while
    it waits - all of them must come back, with exactly their own call's rows (the variable is read once per process:
    a subprocess)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LS_SPIN_CPUS="2")
    out = subprocess.run([sys.executable, "-c", _SLEEPERS.format(root=root)], env=env, capture_output=True, text=True, timeout=300)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")]
    assert line, out.stdout[-2000:] + out.stderr[-2000:]
    bad, batches, requests, parks = (int(x) for x in line[0].split()[1:])
    assert bad == 0
    assert batches >= 1 and requests > batches, (batches, requests)
    assert parks > 0, "no waiter went to sleep although one CPU was allowed to poll"
    print("24 callers, LS_SPIN_CPUS=2:", requests, "requests in", batches, "batches,", parks, "waiters put to sleep")
