"""Scan-path launches that serve several queries per corpus pass (groups of 4 / 8): kernel time per launch
(hipEvents around every launch) and per query, pipelined device API:
    gpurun -- 'python tools/multiq_time.py'      (LEANSEARCH_LIB selects a variant build)"""
import os
import sys; sys.path.insert(0, '/root/repo')
import torch
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H

PIPE = os.environ.get('MQ_PIPE', '1') == '1'  # 0: the selection jobs get their own launch (not in the timed one)
SHAPES = ((200_000, 384, 'f32'), (200_000, 1024, 'f32'), (200_000, 384, 'f16'), (25_000, 384, 'f32'))
if os.environ.get('MQ_SHAPES'):
    SHAPES = tuple(SHAPES[int(i)] for i in os.environ['MQ_SHAPES'].split(','))
for (n, d, dt) in SHAPES:
    c = H.gauss(1234, n, d)
    ix = FlatIPIndex.from_array(c, dtype=dt)
    row = []
    for nq in tuple(int(x) for x in os.environ.get('MQ_NQS', '1,2,4,8,16').split(',')):
        q = torch.from_numpy(H.gauss(5678, nq, d)).cuda()
        for _ in range(30): ix.search_device(q, 50, pipeline=PIPE)
        ix.check()
        ix.set_profiling(True)
        for _ in range(200): ix.search_device(q, 50, pipeline=PIPE)
        ix.check()
        ms, _ = ix.last_kernel_ms()
        ix.set_profiling(False)
        launches = -(-nq // 8) if nq > 4 else 1
        row.append(f"nq={nq}: {ms * 1e3:.1f} us/launch, {ms * 1e3 * launches / nq:.1f} us/query")
    print(f"N={n} d={d} {dt} k=50: " + " | ".join(row), flush=True)
    ix.close()
