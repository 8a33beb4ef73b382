#!/usr/bin/env python
"""bench.py — queries/sec of the dense-retrieval hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c2p|c3|c1|c4]

One "step" = one pass of the hot path (ls_search_device: prep -> scan -> select [-> all-gather
-> merge]) over one batch of synthetic queries, with corpus, queries and outputs resident in HBM.

Workloads (BASELINE.json configs):
  c2  (default) N=200k d=384 fp32, nq=1,    k=50    HBM-bound     <- the headline metric
  c3            N=200k d=384 fp16, nq=1024, k=100   MFMA-bound    <- the metric's "batch=1024" half
  c2p           N=200k d=1024 fp32, nq=1,   k=1000  the reference's real call shape
  c2m           N=1M   d=384 fp32, nq=1,    k=50    c2 with a 1.5 GB corpus (no cache assistance)
  c1            N=10k  d=384 fp32, nq=1,    k=50    the reference's CPU-runnable case
  c4            N=12.5M rows PER GPU (100M at 8 GPUs), d=768 fp16, nq=256, k=100; rows are
                generated on the device per shard (seed 1234+rank); weak scaling

Output (rank 0, stdout), in this order:
  `[bench details] {...}`  everything measured (also written to profiles/bench_last.json): the headline block plus
               "secondary" - the other half of BASELINE's metric (c3), the reference's real call shape (c2p),
               config 4's per-GPU shard (c4), the ls_mq shapes (c2x8, c2px8, c2x32), each with its own roofline /
               recall / parity / cpu_baseline, config 5 (c5) and bm25 at N=1; c3 + the weak-scaled c4 shard at N>1 -
               and "host_api": the synchronous host-array call the reference makes (ls_search, nq=1, PCIe and sync
               inclusive; reference search/engine.py:250) for c2 and c2p, and concurrent callers;
  `[bench summary] {...}`  the same in ~2.5 KB (host_api, bm25, mq, c5);
  the LAST line: ONE JSON line < 2 KB - the contract's keys for the headline workload, with `roofline.batch1024`
               (c3) and `roofline.c4` nested inside `roofline` so that the driver's record carries both halves of
               BASELINE's metric ("batch=1 and 1024").

N > 1: the corpus is row-sharded over the GPUs (strong scaling for c1..c3: the same corpus, the
same queries; the merged top-k is identical to the 1-GPU answer after one RCCL all-gather).
Under a launcher (WORLD_SIZE set: `python -m torch.distributed.run ... bench.py --gpus N`, how the
driver runs it) it is one process per GPU over torch.distributed and the run aborts unless the
communicator really has N ranks on N distinct devices. `python bench.py --gpus N` without a
launcher runs ONE process that drives the N GPUs through the library's own sharded handle
(ls_create_sharded: the reference's process model); `--launcher torchrun` re-executes under
torch.distributed.run instead.
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

WORKLOADS = {
    #        n        d     dtype  nq    k
    "c1": (10_000, 384, "f32", 1, 50),
    "c2": (200_000, 384, "f32", 1, 50),
    "c2p": (200_000, 1024, "f32", 1, 1000),
    "c2m": (1_000_000, 384, "f32", 1, 50),      # c2's shape past the 256 MiB Infinity Cache
    "c2x8": (200_000, 384, "f32", 8, 50),       # 8 queries share one corpus pass (csrc/ls_mq.hip: what combined
    "c2px8": (200_000, 1024, "f32", 8, 1000),   # concurrent callers produce); the reference's shape likewise
    "c2x32": (200_000, 384, "f32", 32, 50),     # 32 queries in ONE exact pass (two MFMA B blocks per A operand, round 6)
    "c3": (200_000, 384, "f16", 1024, 100),
    "c4": (12_500_000, 768, "f16", 256, 100),  # rows PER GPU
}
EXCHANGE_EVERY = 8          # N > 1, batch-1 steps: one all-gather + merge per 8 steps
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F16_PEAK_TF = 2500.0  # dense fp16/bf16 MFMA peak
PREWARM_S = 0.4            # untimed: lets the clocks ramp before warm-up and the timed region


def gauss(seed, n, d):
    x = np.random.default_rng(seed).standard_normal((n, d), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


def algorithmic_bytes(n, d, elem, nq, k):
    """corpus read once per batch + queries in + (f32 score, i64 row) out."""
    return n * d * elem + nq * d * 4 + nq * k * 12


def cpu_baseline(corpus, queries, k, f16, budget_s=14.0):
    """The oracle's vectorised build (same source as the checker, -O3 -mavx2 -mfma, OpenMP) timed
    on this box's host cores on a bounded sample of the same workload. The thread count is the
    best of a short sweep (all cores is rarely the fastest on a many-socket host). Baseline only."""
    from oracle import oracle

    avail = len(os.sched_getaffinity(0))
    try:  # the real FAISS CPU path when the box has it (it does not travel with us)
        import faiss  # type: ignore

        faiss.omp_set_num_threads(avail)
        fx = faiss.IndexFlatIP(corpus.shape[1])
        fx.add(np.ascontiguousarray(corpus, dtype=np.float32))
        qs = np.ascontiguousarray(queries[: min(queries.shape[0], 64)], dtype=np.float32)
        qs = qs[:1] if queries.shape[0] == 1 else qs
        fx.search(qs, k)
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s and done < 5000:
            fx.search(qs, k)
            done += 1
        dt = time.perf_counter() - t0
        out = {"value": round(done * qs.shape[0] / dt, 2), "unit": "queries/s", "cores": avail,
               "kind": "reference",
               "sample": f"faiss.IndexFlatIP (fp32), {done} calls of {qs.shape[0]} queries over "
                         f"the full corpus in {dt:.1f} s, omp threads = {avail}"}
        # what the reference really ships is IndexIVFFlat(nlist = max(256, sqrt(N)))
        # searched with nprobe = 64 (extract/index.py:95-116, search/engine.py:247-248): its rate
        # and its recall against the flat index, on the same arrays
        try:
            n_rows, dim = corpus.shape
            nlist = max(256, int(np.sqrt(n_rows)))
            quant = faiss.IndexFlatIP(dim)
            ivf = faiss.IndexIVFFlat(quant, dim, nlist, faiss.METRIC_INNER_PRODUCT)
            ivf.train(np.ascontiguousarray(corpus[: min(n_rows, 256 * nlist)], dtype=np.float32))
            ivf.add(np.ascontiguousarray(corpus, dtype=np.float32))
            ivf.nprobe = 64
            _, I_flat = fx.search(qs, k)
            _, I_ivf = ivf.search(qs, k)
            hits = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(I_ivf, I_flat))
            done_i, t1 = 0, time.perf_counter()
            while time.perf_counter() - t1 < min(4.0, budget_s / 3) and done_i < 5000:
                ivf.search(qs, k)
                done_i += 1
            dti = time.perf_counter() - t1
            out["ivf"] = {"value": round(done_i * qs.shape[0] / dti, 2), "unit": "queries/s",
                          "nlist": nlist, "nprobe": 64,
                          "recall_vs_flat": round(hits / float(I_flat.size), 5),
                          "what": "faiss.IndexIVFFlat as the reference builds and searches it"}
        except Exception as e:  # the flat leg above is the baseline; this one is informative
            out["ivf"] = {"error": repr(e)}
        return out
    except ImportError:
        pass
    if f16:
        corpus = oracle.c_round_f16(corpus)
    nq = queries.shape[0]
    sample = queries[: min(nq, 64)]

    def rate(threads, seconds, max_calls):
        oracle.set_num_threads(threads)
        oracle.c_search(corpus, sample[:1], k, f16=False, fast=True)  # warm
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds and done < max_calls:
            oracle.c_search(corpus, sample if nq > 1 else sample[:1], k, f16=False, fast=True)
            done += 1
        dt = time.perf_counter() - t0
        return done * (sample.shape[0] if nq > 1 else 1) / dt, done, dt

    cands = sorted({t for t in (8, 16, 32, 64, 128, avail) if t <= avail})
    best_t, best_r = cands[0], 0.0
    for t in cands:
        r, _, _ = rate(t, 0.6, 50)
        if r > best_r:
            best_t, best_r = t, r
    r, done, dt = rate(best_t, max(2.0, budget_s - 0.6 * len(cands)), 5000 if nq == 1 else 200)
    what = (f"{done} single-query searches" if nq == 1
            else f"{done} batches of {sample.shape[0]} of the {nq} queries")
    return {"value": round(r, 2), "unit": "queries/s", "cores": best_t, "kind": "port",
            "sample": f"{what} over the full corpus in {dt:.1f} s; oracle/flat_ip_ref.c built -O3 "
                      f"-mavx2 -mfma with OpenMP, best of {cands} threads on {avail} available "
                      "cores; faiss is not installed on this box"}


class Env:
    """Process-wide facts of this run: the communicator and this rank's device."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.args = torch, dist, args
        # `--gpus N` with no launcher: ONE process drives N GPUs through the library's own sharded
        # handle (ls_create_sharded: per-device streams, RCCL all-gather inside the library).
        # Under a launcher (WORLD_SIZE set, how the driver runs N > 1) it is one process per GPU
        # over torch.distributed.
        self.inlib = args.gpus > 1 and "WORLD_SIZE" not in os.environ
        self.shard_devices = None
        if self.inlib:
            ndev = torch.cuda.device_count()
            self.share_gpu = bool(os.environ.get("LS_BENCH_SHARE_GPU")) and args.gpus > ndev
            if args.gpus > ndev and not self.share_gpu:
                raise SystemExit(f"bench.py: --gpus {args.gpus} but only {ndev} visible GPUs")
            self.shard_devices = [g % max(1, ndev) for g in range(args.gpus)]
            self.world, self.rank, self.local_rank = 1, 0, 0
            self.device_index = self.shard_devices[0]
            torch.cuda.set_device(self.device_index)
            self.dev = torch.device("cuda", self.device_index)
            self.rehearse = False
            self.ranks_seen = args.gpus     # confirmed from the handle after the first exchange
            self.devices_seen = len(set(self.shard_devices))
            return
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}; "
                             "refusing to report a rank count that was not run")
        ndev = torch.cuda.device_count()
        # development rehearsal only: N ranks on fewer devices, collectives over gloo
        self.share_gpu = bool(os.environ.get("LS_BENCH_SHARE_GPU")) and self.world > ndev
        if self.world > ndev and not self.share_gpu:
            raise SystemExit(f"bench.py: {self.world} ranks but only {ndev} visible GPUs")
        self.device_index = self.local_rank % max(1, ndev)
        torch.cuda.set_device(self.device_index)
        self.dev = torch.device("cuda", self.device_index)
        self.rehearse = bool(os.environ.get("LS_BENCH_FORCE_EXCHANGE"))  # 1-GPU rehearsal of N > 1
        self.ranks_seen = 1
        if self.world > 1 or self.rehearse:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            if self.share_gpu:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=self.dev)
            if dist.get_world_size() != args.gpus:
                raise SystemExit(f"bench.py: communicator has {dist.get_world_size()} ranks, "
                                 f"--gpus {args.gpus}")
            # every rank contributes (host, device ordinal) through the collective itself: the
            # count of DISTINCT pairs is what "n_gpus" may claim
            ident = f"{socket.gethostname()}:{self.device_index}:{torch.cuda.get_device_properties(self.dev).name}"
            seen = [None] * self.world
            dist.all_gather_object(seen, ident)
            self.ranks_seen = len(seen)
            self.devices_seen = len(set(seen))
            if self.devices_seen != self.world and not self.share_gpu:
                raise SystemExit(f"bench.py: {self.world} ranks landed on {self.devices_seen} devices")
        else:
            self.devices_seen = 1

    @property
    def n_gpus(self):
        return len(self.shard_devices) if self.inlib else self.world

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def close(self):
        if not self.inlib and (self.world > 1 or self.rehearse):
            self.dist.destroy_process_group()


def run_dense(env: Env, workload: str, steps: int, warmup: int, *, want_cpu: bool, verify: bool,
              c4_rows: int = 0, cpu_budget_s: float = 14.0) -> dict:
    """Build the workload's shard on this rank, time `steps` passes of the hot path, verify the
    timed arrays against the oracle, and return the result block."""
    torch, dist = env.torch, env.dist
    world, rank, dev = env.world, env.rank, env.dev
    from lean_explore_amd.index import FlatIPIndex
    from lean_explore_amd.sharded import ShardedFlatIPIndex, shard_bounds

    n, d, dtype, nq, k = WORKLOADS[workload]
    elem = 2 if dtype == "f16" else 4
    c4 = workload == "c4"
    c4_ref = None
    NV_C4 = 16  # queries of the c4 batch checked against the torch reference
    inlib = env.inlib
    G = env.n_gpus  # shards: ranks under a launcher, devices of the one process otherwise

    def c4_block(seed, rows, device):
        """One shard of config 4 generated in that device's HBM (the 153.6 GB corpus never exists
        on the host): float32 [rows, d], rows L2-normalised."""
        gen = torch.Generator(device=device)
        gen.manual_seed(seed)
        shard = torch.empty((rows, d), dtype=torch.float32, device=device)
        for r0 in range(0, rows, 1 << 20):
            blk = torch.randn((min(1 << 20, rows - r0), d), device=device, generator=gen)
            shard[r0:r0 + blk.shape[0]] = blk / blk.norm(dim=1, keepdim=True)
        return shard

    def c4_reference(shard, q16, row0):
        """torch fp32 reference of the same op on one shard (fp16-rounded operands, fp32
        accumulate), in row blocks: the CPU oracle cannot hold 12.5 M rows."""
        nv = q16.shape[0]
        best_s = torch.full((nv, 0), 0.0, device=shard.device)
        best_i = torch.zeros((nv, 0), dtype=torch.int64, device=shard.device)
        for r0 in range(0, shard.shape[0], 1 << 20):
            sc = q16 @ shard[r0:r0 + (1 << 20)].half().float().T
            ts, ti = sc.topk(min(k, sc.shape[1]), dim=1)
            best_s = torch.cat([best_s, ts], 1)
            best_i = torch.cat([best_i, ti + r0 + row0], 1)
            ts, sel = best_s.topk(min(k, best_s.shape[1]), dim=1)
            best_s, best_i = ts, best_i.gather(1, sel)
        return best_s.cpu(), best_i.cpu()

    if c4:
        # config 4: every GPU generates its own shard (seed 1234 + shard); queries come from one
        # seed, identical everywhere
        rows = c4_rows or n
        n = rows * G
        gen = torch.Generator(device=dev)
        gen.manual_seed(5678)
        tq = torch.randn((nq, d), device=dev, generator=gen)
        tq /= tq.norm(dim=1, keepdim=True)
        queries = tq.cpu().numpy()
        q16 = tq[:NV_C4].half().float()
        if inlib:
            lo, hi = 0, rows
            blocks, refs = [], []
            for g, di in enumerate(env.shard_devices):
                blk = c4_block(1234 + g, rows, torch.device("cuda", di))
                if g == 0:
                    corpus = blk[:200_000].cpu().numpy()  # the CPU baseline's bounded sample
                if verify:
                    refs.append(c4_reference(blk, q16.to(blk.device), g * rows))
                blocks.append(blk)
            local = FlatIPIndex.from_device_blocks(blocks, dtype=dtype)
            del blocks, blk
            if verify:  # merged reference of the whole sharded corpus
                rs, ri = torch.cat([r[0] for r in refs], 1), torch.cat([r[1] for r in refs], 1)
                ts, sel = rs.topk(k, dim=1)
                c4_ref = (ts.numpy(), ri.gather(1, sel).numpy())
        else:
            lo, hi = rank * rows, (rank + 1) * rows
            shard = c4_block(1234 + rank, rows, dev)
            local = FlatIPIndex.from_device_tensor(shard, dtype=dtype, base=lo)
            corpus = shard[:200_000].cpu().numpy()
            if verify:
                rs, ri = c4_reference(shard, q16, lo)
                c4_ref = (rs.numpy(), ri.numpy())
            del shard
        torch.cuda.empty_cache()
    else:
        corpus = gauss(1234, n, d)
        queries = gauss(5678, nq, d)
        if inlib:
            lo, hi = shard_bounds(n, G, 0)
            local = FlatIPIndex.from_array(corpus, dtype=dtype, devices=env.shard_devices)
        else:
            lo, hi = shard_bounds(n, world, rank)
            local = FlatIPIndex.from_array(np.ascontiguousarray(corpus[lo:hi]), dtype=dtype,
                                           device=env.device_index, base=lo)
        tq = torch.from_numpy(queries).to(dev)
    index = None if inlib else ShardedFlatIPIndex(local, n)
    for ov in getattr(env.args, "lib_option", []):
        which, value = ov.split("=")
        local.debug_option(int(which), int(value))
    if inlib and env.share_gpu:
        # rehearsal on fewer GPUs than shards: the per-shard enqueue workers are off by default when
        # shards share a device; force them so the real node's host path is what gets rehearsed
        local.debug_option(11, 1)

    # nq <= 16 is the per-query HBM-bound scan path: consecutive steps are pipelined
    # (LS_FLAG_PIPELINE: launch i = scan of step i + one workgroup finalising step i-1, all on
    # one stream); the last finalize is flushed and everything validated by local.check() inside
    # the timed region.
    # (fp32 index: up to 32 queries are one exact ls_mq pass; fp16 / sharded steps: up to 16)
    scan_nq = 32 if (dtype == "f32" and world == 1 and not inlib) else 16
    pipelined = world == 1 and nq <= scan_nq and not inlib
    #  pipelined scan-path steps may be repaired at local.check() through the output pointers the
    # library kept (up to 256 launches): every step between two checks gets its OWN output buffer, and consecutive
    # steps search DIFFERENT query sets (the same queries rotated by one row per step), so that a repair written to
    # the wrong buffer - or a stale row - shows up in the verification of the last timed steps below
    RING = 256 if pipelined else 64
    NROT = 4 if (pipelined and nq > 1) else 1
    out_ring = [(torch.empty((nq, k), dtype=torch.float32, device=dev),
                 torch.empty((nq, k), dtype=torch.int64, device=dev)) for _ in range(RING)]
    tqs = [tq.roll(r, dims=0).contiguous() for r in range(NROT)]
    step_i = [0]
    tail_outs = []

    # N > 1, small batches: the sharded pipeline (local search of step i carries the finalize of
    # step i-1; the all-gather of step i-1 runs asynchronously; step i-2 is merged)
    sharded_pipe = world > 1 and nq <= 16
    if env.rehearse and not inlib:
        index.force_exchange = True
        sharded_pipe, pipelined = nq <= 16, False

    # 1 GPU, large batches (MFMA path): calls are queued on the index's two internal lanes
    # (LS_FLAG_PIPELINE: consecutive batches alternate between two streams, so one batch's small
    # kernels overlap the other's MFMA pass); the verification flags of up to 16 outstanding
    # batches are checked (and flagged queries repaired) by local.check() inside the timed
    # region, instead of one host round trip per batch
    batched_async = world == 1 and not pipelined and not env.rehearse and not inlib

    def step(profile=False):
        o = out_ring[step_i[0] % RING]
        tq = tqs[step_i[0] % NROT]
        step_i[0] += 1
        if pipelined and step_i[0] % RING == 0:
            local.check()  # (the ring of output buffers wraps: make the steps that used it final first)
        if inlib:
            # one process, every GPU: the sharded handle queues the local searches on its per-device
            # streams, one RCCL all-gather and the merge; batches ride the shards' two lanes
            return local.search_device(tq, k, o[0], o[1], asynchronous=nq <= 16, pipeline=nq > 16)
        if pipelined:
            return local.search_device(tq, k, o[0], o[1], pipeline=True)
        if batched_async:
            # the profiling pass queues the same batches on ONE stream (LS_FLAG_ASYNC): under
            # LS_FLAG_PIPELINE the pass launches of the two lanes overlap on purpose, so a pair of
            # events around one of them also measures its wait for CUs, not the kernel
            return local.search_device(tq, k, o[0], o[1], pipeline=not profile, asynchronous=profile)
        if sharded_pipe:
            return index.search_device_pipelined(tq, k, exchange_every=EXCHANGE_EVERY)
        return index.search_device(tq, k)

    def drain():
        if inlib:
            local.check()
        elif sharded_pipe:
            index.flush()
        elif world > 1 or env.rehearse:
            index.finish()  # looks at the gathered verification flags, repairs + re-exchanges if any
        else:
            local.check()

    # ---- verification on the very arrays AND the very code path that were timed. It runs AFTER
    # the timed region: the oracle's OpenMP / BLAS worker threads keep spinning for a while after
    # a call and were measured to double the host's launch cost of the steps that follow.
    parity = {}

    def verify_fn():
        if not verify:
            step()
            drain()
            return None
        if c4:
            # the full nq-query batch through the timed path (batched MFMA kernels); at N > 1 the
            # merged rows that belong to THIS shard must be exactly this shard's reference rows
            # that survive the merge, so compare on the local result before the exchange
            s, i = local.search_device(tq, k, asynchronous=True)
            local.check()
            rs, ri = c4_ref
            gs, gi = s[:NV_C4].cpu().numpy(), i[:NV_C4].cpu().numpy()
            hit = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(gi, ri))
            if not np.allclose(gs, rs, rtol=0, atol=2e-5):
                raise SystemExit("c4: scores differ from the torch fp32 reference")
            step()
            drain()
            return hit / float(ri.size)
        if rank != 0:
            step()
            drain()
            return None
        from oracle import oracle

        step_i[0] = 0  # (query rotation 0: the oracle's row order)
        s, i = step()
        drain()
        nv = min(nq, scan_nq)
        Dr, Ir = oracle.c_search(corpus, queries[:nv], k, f16=(dtype == "f16"))
        _, _, S = oracle.np_search(corpus, queries[:nv], k, f16=(dtype == "f16"))
        got_s, got_i = s[:nv].cpu().numpy(), i[:nv].cpu().numpy()
        rep = oracle.compare_topk(got_s, got_i, Dr, Ir, S)
        parity.update({"checked_queries": nv, "max_score_err_vs_strict": rep["max_score_err"],
                       "index_mismatches_vs_strict": rep["index_mismatches"],
                       "near_ties_excused": rep["near_ties_excused"]})
        if dtype == "f32" and nq <= scan_nq and world == 1 and not inlib:
            # zero excuse: bit-identical (scores and indices) to the oracle run in the scan kernels' own
            # documented fp32 summation order (oracle/flat_ip_ref.c ORDER_SCAN); raises on any difference
            ko = oracle.compare_kernel_order(got_s, got_i, corpus, queries[:nv], k, orders=("scan",))
            parity.update({"kernel_order": "scan", "kernel_order_mismatches": ko["kernel_order_mismatches"],
                           "bit_identical_queries": ko["kernel_order_queries"]["scan"]})
        # the last timed steps' own buffers: the verified rows, rotated like their queries were
        for rot, ts, ti in tail_outs:
            if not (np.array_equal(ts, np.roll(got_s, rot, axis=0)) and np.array_equal(ti, np.roll(got_i, rot, axis=0))):
                raise SystemExit(f"{workload}: a timed step's output buffer (query rotation {rot}) differs from the verified result")
        if tail_outs:
            parity["timed_steps_rechecked"] = len(tail_outs)
        return rep["recall"]

    # untimed pre-warm: a 20-step driver run is ~1 ms of GPU time, shorter than the clock ramp.
    # Every rank must run the SAME number of steps (they contain collectives): one group is
    # timed, the group count is agreed on by a max-reduction.
    def group():
        for _ in range(8):
            step()
        drain()

    group()
    env.barrier()
    t_pw = time.perf_counter()
    group()
    t_group = max(time.perf_counter() - t_pw, 1e-5)
    n_groups = int(min(400, max(1, PREWARM_S / t_group)))
    if world > 1:
        t = torch.tensor([n_groups], dtype=torch.int64, device="cpu" if env.share_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n_groups = int(t.item())
    for _ in range(n_groups):
        group()
    for _ in range(warmup):
        step()
    env.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    launches0 = local.debug_counter(11)  # (kernel launches the handle has queued so far)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        step()
    ev1.record()
    launches_timed_region = local.debug_counter(11) - launches0
    drain()  # flushes the pipeline and synchronises
    env.barrier()
    dt = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if pipelined and verify:  # the outputs of the last timed steps, before the profiling pass reuses their buffers
        for j in range(min(NROT, steps)):
            i = step_i[0] - 1 - j
            tail_outs.append((i % NROT, out_ring[i % RING][0].cpu().numpy(), out_ring[i % RING][1].cpu().numpy()))
    if world > 1:
        t = torch.tensor([dt, dev_ms], dtype=torch.float64, device=dev)
        if env.share_gpu:
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dev_ms = t.tolist()

    # ---- dominant-kernel duration: hipEvents around EVERY launch of it, on the stream it runs
    # on, over a second pass of the same steps (events would perturb the timed pass)
    local.set_profiling(True)
    n_prof = min(steps, 4096 // max(1, min(nq, scan_nq)))
    for _ in range(n_prof):
        step(profile=True)
    drain()
    env.barrier()
    scan_ms_avg, _total = local.last_kernel_ms()
    local.set_profiling(False)
    recall = verify_fn()
    repaired = local.debug_counter(8) if (nq > 16 and dtype == "f16") else 0
    if inlib:
        # what the handle itself says about the exchange: 2 = RCCL communicators initialised and
        # used; 0 = copies (shards sharing a device cannot form a communicator)
        env.rccl_in_library = local.debug_counter(15) == 2
        env.shards_seen = local.shards()
        # transport, RCCL version / failure text, enqueue workers, hipDeviceCanAccessPeer matrix
        env.exchange_info = local.exchange_info()
    ev_ms = scan_ms_avg
    roof_src = "mean of hipEvent pairs bracketing each launch (second pass of the same steps)"
    if batched_async:
        roof_src = ("mean of hipEvent pairs attached to each MFMA-pass dispatch, second pass of the same "
                    "batches queued on ONE stream (LS_FLAG_ASYNC): in the timed, pipelined run the two "
                    "lanes' pass launches overlap and a launch's duration includes its wait for CUs")
    one_mq_launch_per_step = (pipelined and dtype == "f32" and 2 <= nq <= scan_nq and world == 1
                              and steps <= launches_timed_region <= steps + steps // 128 + 1)
    if pipelined and (nq == 1 or one_mq_launch_per_step):
        # one launch per step, back to back on one stream: the timed region's own hipEvents give
        # the average launch duration (kernel boundary included) without per-launch event overhead
        # (round 6: also for the ls_mq shapes - ONE launch per step, counted by the handle's launch counter over
        # the timed region: steps + the stand-alone selection the library runs by itself every 256 launches, whose
        # time stays inside this average; the per-launch event pairs of the second pass read 4 % above rocprofv3's
        # kernel average there - 49.4 vs 47.3 us at c2x8 -, this reads 47.5: what `ms_per_step` already says)
        scan_ms_avg = dev_ms / steps
        roof_src = ("timed region: hipEvent pair around the K back-to-back launches / K "
                    "(kernel boundary included); event-bracketed mean in kernel_ms_bracketed")
    n_local = hi - lo
    batch_ms = dt * 1e3 / steps
    mfma_path = nq > 16 and dtype == "f16"
    if c4:  # 256 FLOP/B, just under the ridge (312): HBM-bound; the MFMA fraction rides along
        ab = algorithmic_bytes(n_local, d, elem, nq, k)
        ach = ab / (scan_ms_avg * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                "mfma_frac": round(2.0 * nq * n_local * d / (scan_ms_avg * 1e-3) / 1e12
                                   / MFMA_F16_PEAK_TF, 4),
                "frac_whole_batch": round(ab / (batch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    elif workload == "c3":
        flops = 2.0 * nq * n_local * d
        ach = flops / (scan_ms_avg * 1e-3) / 1e12
        roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_F16_PEAK_TF,
                "unit": "TFLOP/s", "frac": round(ach / MFMA_F16_PEAK_TF, 4), "traffic": None,
                # the same flops over the WHOLE batch (prep, sample pass, tau, MFMA pass, select)
                "frac_whole_batch": round(flops / (batch_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TF, 4)}
    else:
        ab = algorithmic_bytes(n_local, d, elem, 1, k)
        ach = ab / (scan_ms_avg * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None}
    mq_path = dtype == "f32" and 2 <= nq <= scan_nq
    roof["kernel"] = "ls_gemm_filter_kernel" if mfma_path else ("ls_mq_kernel" if mq_path else "ls_scan_kernel")
    roof["kernel_ms"] = round(scan_ms_avg, 5)
    roof["kernel_ms_source"] = roof_src
    roof["kernel_ms_bracketed"] = round(ev_ms, 5)
    roof["launches_in_timed_region"] = int(launches_timed_region)
    roof["launches_timed"] = steps if (pipelined and (nq == 1 or one_mq_launch_per_step)) else n_prof * (1 if mq_path else nq if nq <= 16 else 1)
    pmc = ROOT / "profiles" / f"pmc_{workload}.json"
    if pmc.exists():  # HBM bytes per launch from rocprofv3 --pmc (profiles/collect.sh)
        try:
            roof["traffic"] = json.loads(pmc.read_text()).get("hbm_bytes_per_launch")
            roof["traffic_source"] = f"profiles/{pmc.name}"
        except Exception:
            pass
    roof["algorithmic_bytes"] = algorithmic_bytes(n_local, d, elem, 1 if nq <= scan_nq else nq, k)

    res = {
        "value": round(nq * steps / dt, 1),
        "unit": "queries/s",
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": round(batch_ms, 5),
        "device_ms_per_step": round(dev_ms / steps, 5),
        "scaling": "weak" if c4 else "strong",
        "n_shards": G,
        "dtype": dtype,
        "data": ("synthetic (standard-normal rows, L2-normalised; generated in HBM per shard, "
                 "corpus seed 1234+rank, query seed 5678)" if c4 else
                 "synthetic (standard-normal rows, L2-normalised; corpus seed 1234, query seed 5678)"),
        "config": {"workload": f"{workload}: N={n} d={d} {dtype} nq={nq} k={k}",
                   "rows_per_gpu": n_local, "parallelism": f"row-shard x{G}",
                   "process_model": ("one process, sharded handle (ls_create_sharded)" if inlib
                                     else "one process per GPU (torch.distributed)" if world > 1
                                     else ("one process, one GPU; the exchange code path is forced "
                                           "over a one-rank process group (rehearsal)" if env.rehearse
                                           else "one process, one GPU (no process group)")),
                   "exchange": ("one in-library all-gather + merge per step" if inlib else
                                f"pipelined: one packed all-gather + merge per {EXCHANGE_EVERY} "
                                "steps, overlapping the next scans"
                                if sharded_pipe else ("one all-gather per step" if world > 1
                                                      else "none")),
                   "launches_per_step": (1 if pipelined else
                                         round(1 + 2.0 / EXCHANGE_EVERY, 3) if sharded_pipe else
                                         (local.debug_counter(9) if mfma_path else 2 * nq)
                                         + (2 if world > 1 else 0)),
                   "note": (f"each launch = one corpus pass for the {nq} queries on the f32 matrix cores (ls_mq_kernel, "
                            "bit-identical to the single-query scan) + the workgroups finalising step i-1"
                            if (pipelined and mq_path) else
                            "each launch = scan(step i) + one workgroup finalising step i-1")
                   if (pipelined or sharded_pipe) else
                   ("batched MFMA path, pipelined: query prep, MFMA pass (+ the sample phase of the "
                    "batch two calls ahead), tau, select - on two lanes + a select stream"
                    if (mfma_path and batched_async) else
                    "batched MFMA path: query prep, sample pass, tau, MFMA pass, select"
                    if mfma_path else "scan + select launches per query")},
        "effective_gbs": round(algorithmic_bytes(n_local, d, elem, nq, k) * steps / dt / 1e9, 1),
        "recall_at_k": recall,
        # queries the batched path handed to the exact scan path (speculative threshold let
        # < k rows through, or a candidate queue overflowed) over the whole run
        "repaired_queries": repaired,
        "roofline": roof,
    }
    if parity:
        res["parity"] = parity
    if want_cpu and rank == 0 and world == 1 and not inlib:
        cb = cpu_baseline(corpus, queries, k, dtype == "f16", budget_s=cpu_budget_s)
        if c4:  # timed on the first 200k rows; a flat scan is linear in the row count
            cb["value"] = round(cb["value"] * corpus.shape[0] / n, 3)
            cb["sample"] = (f"first {corpus.shape[0]} of the {n} rows, rate scaled by "
                            f"{corpus.shape[0]}/{n}; " + cb["sample"])
        res["cpu_baseline"] = cb
    local.close()
    del index, local, out_ring, tq
    torch.cuda.empty_cache()
    return res


def run_host_api(env: Env, workload: str, calls: int = 300) -> dict:
    """The call the reference actually makes (search/engine.py:238-250): host float32 [1, d] in,
    host (D, I) out, synchronous. Wall-clock per call, PCIe transfers and the stream sync
    included; normalisation fused (LS_FLAG_NORMALIZE) as INTEGRATION.md's patch does."""
    from lean_explore_amd.index import FlatIPIndex

    n, d, dtype, nq, k = WORKLOADS[workload]
    corpus = gauss(1234, n, d)
    q = gauss(5678, 1, d)
    ix = FlatIPIndex.from_array(corpus, dtype=dtype, device=env.device_index)
    del corpus
    for _ in range(30):
        ix.search(q, k, normalize=True)
    lat = np.empty(calls)
    for i in range(calls):
        t0 = time.perf_counter()
        ix.search(q, k, normalize=True)
        lat[i] = time.perf_counter() - t0
    # the same call through the plain ctypes binding (north_star: "thin ctypes shim"); FlatIPIndex.search above
    # goes through csrc/lsfast.c when it is built (the same C symbol without ctypes' argument conversion)
    from lean_explore_amd import native

    lib, h = native.load(), ix._handle
    D = np.empty((1, k), np.float32)
    I = np.empty((1, k), np.int64)
    lat_ct = np.empty(calls)
    for i in range(30 + calls):
        t0 = time.perf_counter()
        rc = lib.ls_search(h, native.addr(q), 1, k, native.LS_FLAG_NORMALIZE, native.addr(D), native.addr(I))
        if i >= 30:
            lat_ct[i - 30] = time.perf_counter() - t0
        if rc:
            native.check(rc)
    # the launch's own duration on THIS box (hipEvent pair around each call's launch, on the stream it runs on: the
    # scan + the selection workgroup riding in it), over a separate loop - the events cost the call ~2 us. Tells a
    # slow box from a regression of the host path: p50 - scan_kernel_us = launch + PCIe + polling overhead.
    scan_us = None
    try:
        ix.set_profiling(True)
        for _ in range(100):
            ix.search(q, k, normalize=True)
        scan_us = round(ix.last_kernel_ms()[0] * 1e3, 2)
        ix.set_profiling(False)
    except Exception:
        pass
    ix.close()
    mean = float(lat.mean())
    return {"workload": f"{workload}: N={n} d={d} {dtype} nq=1 k={k}", "calls": calls,
            "scan_kernel_us": scan_us,
            "binding": "csrc/lsfast.c (CPython)" if native.fast_search() is not None else "ctypes",
            "us_per_call_p50_ctypes_binding": round(float(np.median(lat_ct)) * 1e6, 2),
            "us_per_call_mean": round(mean * 1e6, 2),
            "us_per_call_p50": round(float(np.median(lat)) * 1e6, 2),
            "us_per_call_p90": round(float(np.quantile(lat, 0.9)) * 1e6, 2),
            "queries_per_s": round(1.0 / mean, 1),
            "hbm_frac_of_call": round(algorithmic_bytes(n, d, 4 if dtype == "f32" else 2, 1, k)
                                      / mean / 1e9 / HBM_PEAK_GBS, 4),
            "what": "ls_search(host q[1,d], LS_FLAG_NORMALIZE) -> host (D, I); H2D, kernels, D2H "
                    "and stream sync inside the timed call"}


def run_callers(env: Env, workload: str = "c2", secs: float = 0.6) -> dict:
    """T threads of synchronous single-query ls_search on ONE handle (an MCP server with several clients,
    reference mcp/server.py:147-151 -> search/engine.py:250): queries/s and p50 latency for T = 1, 2, 4, 8, 16.
    Concurrent requests are combined into shared corpus passes (fp32: csrc/ls_mq.hip, bit-identical rows) and
    a leader with nothing in flight waits briefly for the callers seen lately, so that they share ONE pass
    (gather, debug option 20); tools/concurrent_callers.py and tools/callers_c.c A/B the mechanisms."""
    import threading

    from lean_explore_amd.index import FlatIPIndex

    n, d, dtype, _, k = WORKLOADS[workload]
    corpus = gauss(1234, n, d)
    q = gauss(5678, 16, d)
    ix = FlatIPIndex.from_array(corpus, dtype=dtype, device=env.device_index)
    del corpus
    for _ in range(50):
        ix.search(q[:1], k, normalize=True)
    out = {"workload": f"{workload}: N={n} d={d} {dtype} k={k}, one query per call", "seconds_per_point": secs}
    for T in (1, 2, 4, 8, 16):
        counts, lats = [0] * T, [[] for _ in range(T)]
        stop = [0.0]

        def w(t):
            qq = q[t:t + 1]
            while time.perf_counter() < stop[0]:
                t0 = time.perf_counter()
                ix.search(qq, k, normalize=True)
                lats[t].append(time.perf_counter() - t0)
                counts[t] += 1

        th = [threading.Thread(target=w, args=(t,)) for t in range(T)]
        t0 = time.perf_counter()
        stop[0] = t0 + secs
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t0
        allat = np.concatenate([np.asarray(l) for l in lats])
        out[f"callers_{T}"] = {"queries_per_s": round(sum(counts) / dt, 1), "p50_us": round(float(np.median(allat)) * 1e6, 1)}
    out["combined_batches"] = ix.debug_counter(16)
    out["requests_in_combined_batches"] = ix.debug_counter(17)
    out["overlapped_calls"] = ix.debug_counter(24)
    ix.close()
    return out


def run_callers_c() -> dict:
    """The same measurement from C threads (tools/callers_c.c, built with gcc on the spot): what the library's
    caller gathering delivers without the Python threads' GIL hand-offs. One 0.8 s run per point."""
    import re
    import subprocess
    import tempfile

    root = os.path.dirname(os.path.abspath(__file__))
    from lean_explore_amd import native

    lib = os.environ.get("LEANSEARCH_LIB") or os.path.join(root, "lean-explore_amd", "libleansearch.so")
    native.load()  # (fails loudly if the library is missing)
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "callers_c")
        subprocess.run(["gcc", "-O2", os.path.join(root, "tools", "callers_c.c"), "-o", exe, "-ldl", "-lm", "-lpthread"],
                       check=True, capture_output=True, timeout=120)
        txt = subprocess.run([exe, lib, "1", "-1", "1"], check=True, capture_output=True, timeout=300, text=True).stdout
    out = {"harness": "tools/callers_c.c: T pthreads, one synchronous single-query ls_search per call, 0.8 s per point"}
    for m in re.finditer(r"N=(\d+) d=(\d+) k=(\d+),\s+(\d+) callers:\s+(\d+) q/s, p50 ([\d.]+) us", txt):
        n, d, k, T, qps, p50 = m.groups()
        out.setdefault(f"N={n} d={d} k={k}", {})[f"callers_{T}"] = {"queries_per_s": float(qps), "p50_us": float(p50)}
    return out


def run_bm25(n_docs: int = 200_000, k: int = 1000, calls: int = 300) -> dict:
    """BM25+ name retrieval (reference search/engine.py:192-223; bm25s's eager-sparse
    scoring) on the HIP kernels: one 3-token query over `n_docs` synthetic declaration names, synchronous
    host call, checked bit for bit against the numpy oracle outside the timed loop."""
    from lean_explore_amd.bm25 import BM25Index
    from lean_explore_amd.search.tokenization import tokenize_spaced

    words = ("add comm assoc zero one mul le lt succ pred map filter append length nil cons sum prod "
             "continuous measurable integral deriv norm inner dist open closed compact").split()
    rng = np.random.default_rng(3)
    names = []
    for i in range(n_docs):
        parts = [words[j] for j in rng.integers(0, len(words), size=rng.integers(1, 5))]
        ns = ["Nat", "List", "Real", "MeasureTheory", "Mathlib"][rng.integers(0, 5)]
        names.append(f"{ns}.{'_'.join(parts)}{i % 97 if i % 3 == 0 else ''}")
    corpus = [list(dict.fromkeys(tokenize_spaced(nm))) for nm in names]
    ix = BM25Index().index(corpus)
    q = ["nat", "add", "comm"]
    ids = ix.token_ids(q)
    postings = int(sum(int(ix.indptr[t + 1] - ix.indptr[t]) for t in ids))
    nnz = int(ix.indptr[-1])
    for _ in range(20):
        ix.retrieve(q, k)
    lat = np.empty(calls)
    for i in range(calls):
        t0 = time.perf_counter()
        ix.retrieve(q, k)
        lat[i] = time.perf_counter() - t0
    out = {"workload": f"bm25: {n_docs} names, 3-token query, k={k}", "calls": calls,
           "us_per_query_p50": round(float(np.median(lat)) * 1e6, 2),
           "us_per_query_mean": round(float(lat.mean()) * 1e6, 2),
           "queries_per_s": round(1.0 / float(lat.mean()), 1),
           "launches_per_query": 2,
           # what the doc-major score kernel reads per query (document pointers + every (token, value)
           # entry + the score vector it writes) against the three posting lists a CSC walk would touch
           "algorithmic_bytes_doc_major": 4 * (n_docs + 1) + 8 * nnz + 4 * n_docs,
           "posting_list_bytes_csc": 8 * postings,
           "selection_left_fast_path": int(ix.debug_counter(0)),
           "what": "BM25Index.retrieve (ls_bm25_search, synchronous host call): score kernel + selection"}
    pmc = ROOT / "profiles" / "pmc_bm25.json"
    if pmc.exists():
        try:
            out["hbm_bytes_per_launch_pmc"] = json.loads(pmc.read_text()).get("hbm_bytes_per_launch")
            out["kernel_avg_us_rocprof"] = json.loads(pmc.read_text()).get("kernel_avg_us_rocprof")
        except Exception:
            pass
    try:
        from oracle import bm25_ref

        docs, sc = ix.retrieve(q, k)
        dref, sref = bm25_ref.retrieve(bm25_ref.build(corpus), q, k)
        out["bit_exact_vs_oracle"] = bool(np.array_equal(docs, dref) and np.array_equal(sc, sref))
    except Exception as e:  # the oracle is the checker: report, never lose the line
        out["bit_exact_vs_oracle"] = repr(e)
    return out


LINE_MAX = 1900             # bytes of the final stdout line (the driver's record keeps ~2 KB of it verbatim)
DETAILS_PREFIX = "[bench details] "
SUMMARY_PREFIX = "[bench summary] "
DETAILS_FILE = ROOT / "profiles" / "bench_last.json"


def compact_line(full: dict) -> dict:
    """The ONE JSON line the driver parses: the headline keys, `roofline` and `cpu_baseline` of the contract,
    with the other half of BASELINE's metric (batch = 1024: config 3) and config 4's per-GPU shard nested INSIDE
    `roofline` (the driver's record keeps that object whole and drops keys it does not know), each in a dozen
    numbers. Everything else (full secondary blocks, caller tables, exchange info) is in DETAILS_FILE and on the
    `[bench details]` line printed before this one."""
    def num(x, nd=5):
        return round(x, nd) if isinstance(x, float) else x

    def sub(x, *, cpu=True):
        r = x["roofline"]
        o = {"workload": x["config"]["workload"], "value": x["value"], "ms_per_step": x["ms_per_step"],
             "bound": r["bound"], "frac": r["frac"]}
        for k in ("frac_whole_batch", "mfma_frac"):
            if k in r:
                o[k] = r[k]
        o.update({"kernel_ms": r["kernel_ms"], "traffic": r.get("traffic"), "recall": x["recall_at_k"]})
        if cpu and "cpu_baseline" in x:
            o["cpu_baseline_qps"] = x["cpu_baseline"]["value"]
        return o

    r = full["roofline"]
    roof = {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms",
                              "launches_timed", "algorithmic_bytes") if k in r}
    sec = full.get("secondary", {})
    if "c3" in sec and "roofline" in sec["c3"]:
        roof["batch1024"] = sub(sec["c3"])
    if "c4" in sec and "roofline" in sec["c4"]:
        roof["c4"] = sub(sec["c4"], cpu=False)
    extra = {}
    for w in ("c2p", "c2m", "c2x8", "c2px8", "c2x32", "c2px32"):  # frac of HBM peak, kernel us, queries/s
        if w in sec and "roofline" in sec[w]:
            extra[w] = [sec[w]["roofline"]["frac"], num(sec[w]["roofline"]["kernel_ms"] * 1e3, 2), sec[w]["value"]]
    if extra:
        roof["other_hbm"] = {"_": "[frac, kernel_us, queries/s]", **extra}
    cfg = full["config"]
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                "higher_is_better", "scaling", "vs_baseline", "dtype")}
    out["metric"] = "queries/sec (exact IP top-k; recall vs FAISS-flat restatement)"
    out["data"] = "synthetic (unit-norm gaussian rows; seeds 1234/5678)"
    out["config"] = {k: cfg[k] for k in ("workload", "rows_per_gpu", "parallelism", "launches_per_step") if k in cfg}
    out["recall_at_k"] = full.get("recall_at_k")
    if full["n_gpus"] > 1:
        out["rccl_ranks_seen"], out["devices_seen"] = full.get("rccl_ranks_seen"), full.get("devices_seen")
        if "exchange" in full:
            out["exchange"] = str(full["exchange"])[:80]
    if "rehearsal" in full:
        out["rehearsal"] = "shards share GPUs: code path only, the value is not a result"
    out["roofline"] = roof
    if "cpu_baseline" in full:
        cb = full["cpu_baseline"]
        out["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                               "sample": cb["sample"][:96]}
    if "parity" in full:
        p = full["parity"]
        out["parity"] = {k: p[k] for k in ("checked_queries", "index_mismatches_vs_strict", "near_ties_excused",
                                           "kernel_order_mismatches") if k in p}
    out["details"] = "'[bench summary]' line above (host_api, bm25, mq, c5); profiles/bench_last.json"
    if len(json.dumps(out)) > LINE_MAX:
        roof.pop("other_hbm", None)
    return out


def summary_line(full: dict) -> dict:
    """Second-to-last stdout line (`[bench summary] {...}`, <= ~3 KB, inside the stdout tail the driver keeps):
    the synchronous host call, concurrent callers, the ls_mq shapes, BM25 and config 5 in a few numbers each."""
    sec, ha = full.get("secondary", {}), full.get("host_api") or {}
    out = {}
    for w in ("c2p", "c2m", "c2x8", "c2px8", "c2x32", "c2px32"):
        x = sec.get(w)
        if x and "roofline" in x:
            r = x["roofline"]
            out[w] = {"workload": x["config"]["workload"], "value": x["value"], "ms_per_step": x["ms_per_step"],
                      "kernel": r["kernel"], "kernel_ms": r["kernel_ms"], "frac": r["frac"], "traffic": r.get("traffic"),
                      "recall": x["recall_at_k"], "kernel_order_mismatches": x.get("parity", {}).get("kernel_order_mismatches")}
    hb = {}
    for w in ("c2", "c2p"):
        if w in ha and "us_per_call_p50" in ha[w]:
            hb[w] = {k: ha[w].get(k) for k in ("us_per_call_p50", "us_per_call_mean", "us_per_call_p90",
                                               "us_per_call_p50_ctypes_binding", "scan_kernel_us", "hbm_frac_of_call")}
    for key, name in (("concurrent_callers", "py_threads_c2"), ("concurrent_callers_c2p", "py_threads_c2p")):
        cc = ha.get(key) or {}
        if "callers_1" in cc:
            hb[name] = {str(T): [cc[f"callers_{T}"]["queries_per_s"], cc[f"callers_{T}"]["p50_us"]] for T in (1, 2, 4, 8, 16)}
    for shape, tab in (ha.get("concurrent_callers_c_threads") or {}).items():
        if isinstance(tab, dict) and "callers_1" in tab:
            hb["c_threads " + shape] = {T[8:]: [v["queries_per_s"], v["p50_us"]] for T, v in tab.items()}
    if hb:
        hb["_"] = "caller tables: callers -> [queries/s, p50 us]"
        out["host_api"] = hb
    bm = sec.get("bm25") or {}
    if "us_per_query_p50" in bm:
        out["bm25"] = {k: bm.get(k) for k in ("workload", "us_per_query_p50", "queries_per_s", "launches_per_query",
                                              "selection_left_fast_path", "bit_exact_vs_oracle", "algorithmic_bytes_doc_major",
                                              "posting_list_bytes_csc", "hbm_bytes_per_launch_pmc")}
    c5 = sec.get("c5") or {}
    if "value" in c5:
        out["c5"] = {k: c5.get(k) for k in ("value", "end_to_end_ms_per_query", "dense_share_of_end_to_end")}
    elif c5:
        out["c5"] = c5
    return out


def emit(full: dict, args) -> None:
    """Rank 0: everything measured -> DETAILS_FILE and one prefixed stdout line (not a JSON line on purpose: the
    driver must never mistake it for the result), then the compact result as the LAST stdout line."""
    try:
        DETAILS_FILE.write_text(json.dumps(full, indent=1) + "\n")
    except OSError:
        pass
    print(DETAILS_PREFIX + json.dumps(full), flush=True)
    summ = summary_line(full)
    if summ:
        print(SUMMARY_PREFIX + json.dumps(summ), flush=True)
    line = json.dumps(compact_line(full))
    assert len(line) <= LINE_MAX + 200, len(line)
    print(line, flush=True)


def self_spawn(args) -> int:
    """`python bench.py --gpus N` with no launcher: run N ranks under torch.distributed.run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port",
           str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--secondary", default="auto",
                    help="auto | none | comma list of workloads reported under 'secondary'")
    ap.add_argument("--no-host-api", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--c4-rows", type=int, default=0, help="rows per GPU for c4 (default 12.5M)")
    ap.add_argument("--lib-option", action="append", default=[], metavar="WHICH=VALUE",
                    help="developer A/B: ls_debug_option(which, value) on every index the run builds")
    ap.add_argument("--launcher", default="inlib", choices=["inlib", "torchrun"],
                    help="--gpus N without a launcher: 'inlib' = one process drives the N GPUs through "
                         "the library's sharded handle; 'torchrun' = re-execute under "
                         "torch.distributed.run, one process per GPU")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and args.launcher == "torchrun":
        sys.exit(self_spawn(args))

    env = Env(args)
    head = run_dense(env, args.workload, args.steps, args.warmup,
                     want_cpu=not args.no_cpu_baseline, verify=not args.no_verify,
                     c4_rows=args.c4_rows)
    if args.secondary == "auto":
        sec = []
        if args.workload == "c2":  # the default, driver-timed run carries both halves of the metric,
            # the reference's real call shape and the per-GPU shard of config 4
            # ... and (1 GPU) config 2's shape at N = 1 M rows: 1.5 GB, six times the Infinity Cache
            # ... and the small-batch shapes (8 queries per corpus pass on the f32 matrix cores)
            sec = ["c3", "c2p", "c2m", "c2x8", "c2px8", "c2x32", "c4"] if env.n_gpus == 1 else ["c3", "c4"]
    elif args.secondary in ("none", ""):
        sec = []
    else:
        sec = [w for w in args.secondary.split(",") if w in WORKLOADS and w != args.workload]
    secondary = {}
    for w in sec:
        nq = WORKLOADS[w][3]
        # (c3: 1000 pipelined batches = 0.15 s: the chain's fill and drain and its per-128-calls check are amortised)
        st, wu = (1000, 50) if (nq <= 16 or w == "c2x32") else ((1000, 50) if w == "c3" else (30, 3))
        secondary[w] = run_dense(env, w, st, wu, want_cpu=not args.no_cpu_baseline and w not in ("c2m", "c2x8", "c2px8", "c2x32"),
                                 verify=not args.no_verify, c4_rows=args.c4_rows, cpu_budget_s=9.0)
    if (args.secondary == "auto" and args.workload == "c2" and env.n_gpus == 1) or \
            "c5" in args.secondary.split(","):
        # BASELINE config 5: the full hybrid pipeline through Service.search() (embed on PyTorch-ROCm
        # -> BM25 names + dense top-1000 on the HIP kernels -> RRF -> dependency boost -> cross-encoder
        # rerank on PyTorch-ROCm), end-to-end queries/s with the dense stage's share
        try:
            from tools import hybrid_bench

            secondary["c5"] = hybrid_bench.run(queries=12)
        except Exception as e:  # the model stack is plumbing around the path: never lose the line to it
            secondary["c5"] = {"error": repr(e)}
    if args.secondary == "auto" and args.workload == "c2" and env.n_gpus == 1 and env.rank == 0:
        try:
            secondary["bm25"] = run_bm25()
        except Exception as e:
            secondary["bm25"] = {"error": repr(e)}
    host_api = None
    if not args.no_host_api and env.n_gpus == 1 and args.workload == "c2" and env.rank == 0:
        host_api = {w: run_host_api(env, w) for w in ("c2", "c2p")}
        try:
            host_api["concurrent_callers"] = run_callers(env, "c2")
            host_api["concurrent_callers_c2p"] = run_callers(env, "c2p", secs=0.5)  # (the reference's call shape)
        except Exception as e:
            host_api["concurrent_callers"] = {"error": repr(e)}
        try:
            host_api["concurrent_callers_c_threads"] = run_callers_c()
        except Exception as e:
            host_api["concurrent_callers_c_threads"] = {"error": repr(e)}

    if env.rank == 0:
        out = {
            "metric": "queries/sec (exact inner-product top-k, recall vs FAISS-flat restatement)",
            "value": head["value"],
            "unit": head["unit"],
            "n_gpus": env.n_gpus,
            "rccl_ranks_seen": ((env.n_gpus if getattr(env, "rccl_in_library", False) else 0)
                                if env.inlib else env.ranks_seen),
            "devices_seen": (len({sh[0] for sh in getattr(env, "shards_seen", [])})
                             if env.inlib else env.devices_seen),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "device_ms_per_step": head["device_ms_per_step"],
            "higher_is_better": True,
            "scaling": head["scaling"],
            "vs_baseline": None,
            "dtype": head["dtype"],
            "data": head["data"],
            "config": head["config"],
            "effective_gbs": head["effective_gbs"],
            "recall_at_k": head["recall_at_k"],
            "repaired_queries": head["repaired_queries"],
            "prewarm_s": PREWARM_S,
            "roofline": head["roofline"],
        }
        if "parity" in head:
            out["parity"] = head["parity"]
        if "cpu_baseline" in head:
            out["cpu_baseline"] = head["cpu_baseline"]
        if secondary:
            out["secondary"] = secondary
        if host_api:
            out["host_api"] = host_api
        if env.inlib:
            info = getattr(env, "exchange_info", {}) or {}
            exch = info.get("exchange", "unknown")
            if info.get("rccl_error"):  # RCCL could not be used on this node: the handle fell back
                exch += f" ({info['rccl_error']})"
            out["process_model"] = ("one process: ls_create_sharded over devices "
                                    f"{env.shard_devices}; exchange = {exch}")
            out["exchange"] = exch
            out["exchange_info"] = info
            if not env.share_gpu and out["devices_seen"] != env.n_gpus:
                raise SystemExit("bench.py: the sharded handle did not land on N distinct devices")
        if env.share_gpu:
            out["rehearsal"] = (f"{env.n_gpus} shards share {out['devices_seen']} GPU(s): "
                                "exercises the N > 1 code path only, the value is not a result")
        emit(out, args)
    env.close()


if __name__ == "__main__":
    main()
