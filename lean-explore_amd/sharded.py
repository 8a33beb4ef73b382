"""Row-sharded exact search across the GPUs of one node.

The reference is single-process and single-device (its index lives in host RAM,
reference src/lean_explore/search/engine.py:159); sharding is this build's addition:

    rank g holds rows [g*ceil(N/G), min(N, (g+1)*ceil(N/G)))        (contiguous row blocks)
    1. every rank runs the same HIP search on its shard; ls_set_base makes the returned row
       indices global
    2. ONE exchange step: all-gather of the per-rank (scores f32, rows i64) [nq, k]
       (torch.distributed, backend "nccl" == RCCL over xGMI; messages are tiny, latency-bound)
    3. every rank merges the G sorted lists with the HIP merge kernel (ls_merge_topk) under the
       same total order (score desc, global row asc)  ->  identical to the 1-GPU result, bit for
       bit, for every G.

One process per GPU, launched by torchrun. (The same partitioning inside ONE process, behind the C
ABI, is ls_create_sharded / ``FlatIPIndex(devices=[...])``: csrc/ls_shard.hip.) ``local_search`` / ``merge`` are injectable so that
the collective plumbing is testable with gloo on CPU-only hosts (tests/test_sharded_cpu.py
injects the CPU oracle there; the defaults below are the HIP kernels and nothing else).
"""

from __future__ import annotations

from typing import Callable

import numpy as np


def shard_bounds(n: int, world: int, rank: int) -> tuple[int, int]:
    """Rows [lo, hi) held by ``rank``: contiguous blocks of ceil(n / world) rows."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    per = -(-n // world) if n else 0
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi


class ShardedFlatIPIndex:
    """This rank's shard plus the exchange + merge steps."""

    def __init__(self, local_index, n_total: int, *, group=None,
                 local_search: Callable | None = None, merge: Callable | None = None):
        import torch.distributed as dist

        self.local = local_index
        self.n_total = int(n_total)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._local_search = local_search or self._hip_local_search
        self._merge = merge or self._hip_merge
        # testing hook: run the exchange + merge even with a single rank (exercises the RCCL
        # collective on a one-GPU box)
        self.force_exchange = False
        self._pipe = None
        self._unfinished: list = []  # batched exchanges whose flags have not been looked at yet

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_array(cls, corpus: np.ndarray, dtype="f32", device: int = 0, group=None
                   ) -> "ShardedFlatIPIndex":
        """Every rank passes the same full host corpus; each keeps only its row block."""
        import torch.distributed as dist

        from .index import FlatIPIndex

        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        lo, hi = shard_bounds(corpus.shape[0], world, rank)
        local = FlatIPIndex.from_array(np.ascontiguousarray(corpus[lo:hi]), dtype=dtype,
                                       device=device, base=lo)
        return cls(local, corpus.shape[0], group=group)

    @property
    def ntotal(self) -> int:
        return self.n_total

    @property
    def d(self) -> int:
        return self.local.d

    # ------------------------------------------------------------------ HIP defaults
    def _hip_local_search(self, q, k, normalize, out_scores=None, out_indices=None):
        # Always queued asynchronously. The per-query scan path is exact in stream order; the
        # batched MFMA path may flag queries for repair, and those flags travel WITH the results
        # (see search_device): no host synchronisation before the exchange.
        return self.local.search_device(q, k, out_scores, out_indices, normalize=normalize,
                                        asynchronous=True)

    def _hip_merge(self, all_scores, all_rows, k, list_stride_bytes=None):
        import torch

        from . import native

        g, nq = all_scores.shape[0], all_scores.shape[1]
        out_s = torch.empty((nq, k), dtype=torch.float32, device=all_scores.device)
        out_i = torch.empty((nq, k), dtype=torch.int64, device=all_scores.device)
        dev = all_scores.device.index or 0
        stream = torch.cuda.current_stream(all_scores.device).cuda_stream
        lib = native.load()
        if list_stride_bytes is None:
            native.check(lib.ls_merge_topk(all_scores.data_ptr(), all_rows.data_ptr(), g, nq, k,
                                           out_s.data_ptr(), out_i.data_ptr(), dev, stream))
        else:
            native.check(lib.ls_merge_topk_strided(all_scores.data_ptr(), all_rows.data_ptr(),
                                                   list_stride_bytes, g, nq, k, out_s.data_ptr(),
                                                   out_i.data_ptr(), dev, stream))
        return out_s, out_i

    # ------------------------------------------------------------------ search
    def search_device(self, q, k: int, *, normalize: bool = False):
        """q: float32 tensor [nq, d] (same on every rank). Returns (scores, rows) [nq, k]
        tensors, identical on every rank. All work is queued on the current stream and the local
        search is ALWAYS asynchronous: for any batched call (the shards' speculative MFMA paths:
        nq > 16 on fp16 shards, nq > 32 on fp32 shards) the tensors are provisional until
        :meth:`finish`, with one rank as with many (``finish`` then is the local ls_check)."""
        import torch
        import torch.distributed as dist

        if self.world == 1 and not self.force_exchange:
            return self._local_search(q, k, normalize)
        if len(self._unfinished) >= 8:  # bounded look-ahead (the library itself allows 16)
            self.finish()
        nq = q.shape[0]
        hip = self._local_search == self._hip_local_search
        # one packed block per rank: [scores f32 nq*k | pad to 8 B | rows i64 nq*k | flags u32 nq]
        # -> ONE all-gather. The flags are the shard's per-query "provisional" marks.
        sbytes = (nq * k * 4 + 7) & ~7
        fbytes = nq * 4 if hip else 0
        block = (sbytes + nq * k * 8 + fbytes + 7) & ~7
        packed = torch.empty(block, dtype=torch.uint8, device=q.device)
        s_loc = packed[: nq * k * 4].view(torch.float32).view(nq, k)
        i_loc = packed[sbytes: sbytes + nq * k * 8].view(torch.int64).view(nq, k)
        if hip:
            self._hip_local_search(q, k, normalize, s_loc, i_loc)
            self.local.export_flags(packed[sbytes + nq * k * 8: sbytes + nq * k * 8 + fbytes]
                                    .view(torch.int32))
        else:  # injected (CPU tests): copy its result into the packed block
            s_tmp, i_tmp = self._local_search(q, k, normalize)
            s_loc.copy_(s_tmp)
            i_loc.copy_(i_tmp)
        gathered = torch.empty(self.world * block, dtype=torch.uint8, device=q.device)
        dist.all_gather_into_tensor(gathered, packed, group=self.group)
        if self._merge == self._hip_merge:
            # strided views are never materialised: the kernel walks the packed blocks
            out = self._merge_packed(gathered, sbytes, block, nq, k)
        else:
            g2 = gathered.view(self.world, block)
            s_all = g2[:, : nq * k * 4].contiguous().view(torch.float32).view(self.world, nq, k)
            i_all = g2[:, sbytes: sbytes + nq * k * 8].contiguous().view(torch.int64) \
                .view(self.world, nq, k)
            out = self._merge(s_all, i_all, k)
        if hip:  # `packed` holds the local results: it must outlive the ls_check that may repair them
            self._unfinished.append((nq, k, out, packed, gathered, sbytes, block, fbytes))
        return out

    def finish(self):
        """Make the results of the search_device calls since the last finish() final. Every rank
        reads the SAME gathered flags, so all ranks agree, without another collective, on whether
        any shard had a provisional query; if so (rare) the flagged shard repairs it (ls_check)
        and that batch's exchange + merge is redone into the same output tensors."""
        import torch
        import torch.distributed as dist

        todo, self._unfinished = self._unfinished, []
        if not todo:
            self.local.check()
            return
        marks = [t[4].view(self.world, t[6])[:, t[5] + t[0] * t[1] * 8: t[5] + t[0] * t[1] * 8 + t[7]]
                 .any() for t in todo]
        redo = torch.stack(marks).cpu().tolist()  # synchronises the stream
        self.local.check()  # repairs this shard's flagged rows of the `packed` blocks in place
        for again, (nq, k, out, packed, gathered, sbytes, block, fbytes) in zip(redo, todo):
            if not again:
                continue
            packed[sbytes + nq * k * 8:].zero_()
            dist.all_gather_into_tensor(gathered, packed, group=self.group)
            s2, i2 = self._merge_packed(gathered, sbytes, block, nq, k)
            out[0].copy_(s2)
            out[1].copy_(i2)
        if any(redo):
            torch.cuda.synchronize()

    def _merge_packed(self, gathered, sbytes, block, nq, k):
        import torch

        from . import native

        out_s = torch.empty((nq, k), dtype=torch.float32, device=gathered.device)
        out_i = torch.empty((nq, k), dtype=torch.int64, device=gathered.device)
        dev = gathered.device.index or 0
        native.check(native.load().ls_merge_topk_strided(
            gathered.data_ptr(), gathered.data_ptr() + sbytes, block, self.world, nq, k,
            out_s.data_ptr(), out_i.data_ptr(), dev,
            torch.cuda.current_stream(gathered.device).cuda_stream))
        return out_s, out_i

    # ------------------------------------------------------------------ pipelined search
    # Throughput mode for streams of small query batches (the scan path, nq <= 16). Steps are
    # grouped M = exchange_every at a time; a group shares ONE packed buffer
    # [scores f32 M*nq*k | pad | rows i64 M*nq*k], ONE all-gather and ONE merge launch
    # (fewer, larger collectives: per step only the local scan launch remains).
    #   step i : local HIP search with LS_FLAG_PIPELINE into slot i % M of group i // M; its launch
    #            also finalises step i-1 (stream order), so when the first step of group g+1 has
    #            been queued, group g's packed results are complete;
    #   then   : the all-gather of group g is started asynchronously (backend's own stream, under
    #            the next scans) and group g-1 is waited for and merged.
    # Results of a step are valid after `flush()`; `depth` groups are kept in a ring.
    def search_device_pipelined(self, q, k: int, *, normalize: bool = False, depth: int = 4,
                                exchange_every: int = 8):
        import torch

        nq = q.shape[0]
        if nq > 16:
            raise ValueError("pipelined search is for the scan path (nq <= 16)")
        M = max(1, int(exchange_every))
        p = self._pipe
        stream = torch.cuda.current_stream(q.device).cuda_stream
        key = (nq, k, depth, M, q.device, bool(normalize), stream)
        if p is None or p["key"] != key:
            if p is not None:
                self.flush()
            if not (q.dtype == torch.float32 and q.dim() == 2 and q.is_contiguous()
                    and q.shape[1] == self.local.d):
                raise ValueError("expected a contiguous float32 CUDA tensor [nq, d]")
            from . import native

            dev = q.device
            sbytes = (M * nq * k * 4 + 7) & ~7
            block = sbytes + M * nq * k * 8
            p = self._pipe = {
                "key": key, "i": 0, "sbytes": sbytes, "block": block, "M": M, "nq": nq, "k": k,
                "packed": [torch.empty(block, dtype=torch.uint8, device=dev) for _ in range(depth)],
                "gathered": [torch.empty(self.world * block, dtype=torch.uint8, device=dev)
                             for _ in range(depth)],
                "out": [(torch.empty((M * nq, k), dtype=torch.float32, device=dev),
                         torch.empty((M * nq, k), dtype=torch.int64, device=dev))
                        for _ in range(depth)],
                "work": [None] * depth,
                # the per-step host path is a bare ctypes call on cached addresses: small shards
                # are bound by the host's launch rate, not by the GPU
                "call": native.load().ls_search_device, "check": native.check,
                "handle": self.local._ensure_built(), "stream": stream,
                # (INORDER: the exchange consumes the local results before any ls_check could repair them)
                "flags": native.LS_FLAG_PIPELINE | native.LS_FLAG_INORDER | (native.LS_FLAG_NORMALIZE if normalize else 0),
            }
            slots = []
            for g in range(depth):
                base = p["packed"][g].data_ptr()
                os_, oi_ = p["out"][g]
                slots.append([(base + j * nq * k * 4, base + sbytes + j * nq * k * 8,
                               os_[j * nq:(j + 1) * nq], oi_[j * nq:(j + 1) * nq])
                              for j in range(M)])
            p["slots"] = slots
        i = p["i"]
        grp, j = (i // M) % depth, i % M
        s_ptr, i_ptr, out_s, out_i = p["slots"][grp][j]
        p["check"](p["call"](p["handle"], q.data_ptr(), nq, k, p["flags"], s_ptr, i_ptr,
                             p["stream"]))
        if j == 0 and i >= M:  # the launch above finalised the last step of the previous group
            self._start_exchange((i // M - 1) % depth)
            if i >= 2 * M:
                self._finish_exchange((i // M - 2) % depth, M)
        p["i"] = i + 1
        return out_s, out_i

    def _start_exchange(self, grp: int) -> None:
        import torch.distributed as dist

        p = self._pipe
        if self.world == 1 and not self.force_exchange:
            p["work"][grp] = "local"
            return
        p["work"][grp] = dist.all_gather_into_tensor(p["gathered"][grp], p["packed"][grp],
                                                     group=self.group, async_op=True)

    def _finish_exchange(self, grp: int, steps: int) -> None:
        """Wait for group `grp`'s all-gather and merge its first `steps` steps."""
        import torch

        from . import native

        p = self._pipe
        work = p["work"][grp]
        if work is None:
            return
        M, nq, k = p["M"], p["nq"], p["k"]
        out_s, out_i = p["out"][grp]
        if work == "local":
            packed = p["packed"][grp]
            out_s.copy_(packed[: M * nq * k * 4].view(torch.float32).view(M * nq, k))
            out_i.copy_(packed[p["sbytes"]:].view(torch.int64).view(M * nq, k))
        else:
            work.wait()  # the current stream waits for the collective
            g = p["gathered"][grp]
            native.check(native.load().ls_merge_topk_strided(
                g.data_ptr(), g.data_ptr() + p["sbytes"], p["block"], self.world, steps * nq, k,
                out_s.data_ptr(), out_i.data_ptr(), g.device.index or 0,
                torch.cuda.current_stream(g.device).cuda_stream))
        p["work"][grp] = None

    def flush(self) -> None:
        """Drain the pipeline: the last step's finalize, the outstanding exchanges and merges."""
        import torch

        p = self._pipe
        self.local.check()  # launches the pending finalize and synchronises the stream
        if p is None or p["i"] == 0:
            return
        i, M, depth = p["i"], p["M"], len(p["packed"])
        last = (i - 1) // M  # group of the newest step; it holds (i - 1) % M + 1 steps
        if last >= 1:
            self._finish_exchange((last - 1) % depth, M)  # started, not yet merged
        self._start_exchange(last % depth)
        self._finish_exchange(last % depth, (i - 1) % M + 1)
        torch.cuda.synchronize()
        p["i"] = 0

    def search(self, x: np.ndarray, k: int, *, normalize: bool = False
               ) -> tuple[np.ndarray, np.ndarray]:
        """index.search(x, k) on host arrays (reference search/engine.py:250 signature)."""
        import torch

        x = np.ascontiguousarray(x, dtype=np.float32)
        dev = torch.device("cuda", self.local.device)
        s, i = self.search_device(torch.from_numpy(x).to(dev), int(k), normalize=normalize)
        self.finish()
        return s.cpu().numpy(), i.cpu().numpy()
