#!/usr/bin/env python
"""Pipelined batch-1 steps on a small shard, (historical: argument 2 toggled the two-stage finalize experiment, debug option 8, since removed): step time
here, kernel durations when run under rocprofv3 --kernel-trace --stats.
  python tools/small_n_kernel.py N two_stage(0|1)"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lean_explore_amd.index import FlatIPIndex  # noqa: E402
from tests import helpers as H  # noqa: E402

n, two = int(sys.argv[1]), int(sys.argv[2])
c = H.gauss(1234, n, 384)
tq = torch.from_numpy(H.gauss(5678, 1, 384)).cuda()
ix = FlatIPIndex.from_array(c)
# ix.debug_option(8, two)  # the experiment this script measured was removed
outs = [(torch.empty((1, 50), device="cuda"), torch.empty((1, 50), dtype=torch.int64, device="cuda"))
        for _ in range(16)]
for i in range(500):
    ix.search_device(tq, 50, *outs[i & 15], pipeline=True)
ix.check()
t0 = time.perf_counter()
R = 20000
for i in range(R):
    ix.search_device(tq, 50, *outs[i & 15], pipeline=True)
th = time.perf_counter() - t0
ix.check()
dt = time.perf_counter() - t0
print(f"N={n} two_stage={two}: {dt / R * 1e6:.2f} us/step, host issue {th / R * 1e6:.2f} us/call", flush=True)

# the same steps through a bare ctypes call on cached addresses (what lean_explore_amd.sharded's
# pipelined path does): how much of the step is the Python wrapper?
from lean_explore_amd import native  # noqa: E402

lib, h = native.load(), ix._ensure_built()
call = lib.ls_search_device
st = torch.cuda.current_stream().cuda_stream
args = [(h, tq.data_ptr(), 1, 50, native.LS_FLAG_PIPELINE, o[0].data_ptr(), o[1].data_ptr(), st) for o in outs]
for i in range(500):
    call(*args[i & 15])
ix.check()
t0 = time.perf_counter()
for i in range(R):
    call(*args[i & 15])
th = time.perf_counter() - t0
ix.check()
dt = time.perf_counter() - t0
print(f"N={n} bare ctypes: {dt / R * 1e6:.2f} us/step, host issue {th / R * 1e6:.2f} us/call", flush=True)
