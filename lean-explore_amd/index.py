"""FlatIPIndex — the faiss-shaped object the local search backend holds.

It offers exactly the members the reference uses on its FAISS index
(reference src/lean_explore/search/engine.py:247-250 and tests/extract/index_test.py:172-173):
``search(x, k) -> (D, I)``, ``ntotal``, ``d``, ``add(x)``; it has no ``nprobe`` attribute (the
reference sets it only ``if hasattr(index, "nprobe")``, engine.py:247). All arithmetic happens in
libleansearch.so on the MI355X; this class only owns the handle and marshals numpy / torch
buffers across the C ABI.
"""

from __future__ import annotations

import ctypes
from typing import Any

import numpy as np

from . import native

_DTYPES = {"f32": native.LS_DTYPE_F32, "float32": native.LS_DTYPE_F32,
           "f16": native.LS_DTYPE_F16, "float16": native.LS_DTYPE_F16}


def _dtype_code(dtype: Any) -> int:
    if isinstance(dtype, int):
        return dtype
    try:
        return _DTYPES[str(np.dtype(dtype)) if not isinstance(dtype, str) else dtype]
    except (KeyError, TypeError):
        raise ValueError(f"unsupported storage dtype {dtype!r} (use 'f32' or 'f16')") from None


def normalize_L2(x: np.ndarray, device: int = 0) -> None:
    """In-place row L2 normalisation of a C-contiguous float32 [nq, d] array.

    Counterpart of ``faiss.normalize_L2`` (reference search/engine.py:242): returns None,
    rows with zero norm are left unchanged.
    """
    if not isinstance(x, np.ndarray) or x.dtype != np.float32 or x.ndim != 2 \
            or not x.flags.c_contiguous:
        raise ValueError("normalize_L2 expects a C-contiguous float32 array of shape [nq, d]")
    lib = native.load()
    native.check(lib.ls_normalize_l2(x.ctypes.data, x.shape[0], x.shape[1], device))


def _device_list(devices: Any) -> list[int] | None:
    """``None`` -> single device; ``"all"`` -> every visible GPU; else a list of ordinals."""
    if devices is None:
        return None
    if isinstance(devices, str):
        if devices.strip().lower() == "all":
            return list(range(native.device_count()))
        devices = [int(t) for t in devices.replace(",", " ").split()]
    out = [int(t) for t in devices]
    if not out:
        raise ValueError("devices must name at least one GPU")
    return out


class FlatIPIndex:
    """Exact inner-product index resident in HBM: one GPU, or — ``devices=[...]`` — row-sharded
    over several GPUs of the node inside this one process (ls_create_sharded: concurrent local
    top-k, one RCCL all-gather of the packed results, merge on ``devices[0]``; bit-identical
    to the single-GPU answer). The reference's backend is one process holding one index
    (reference mcp/server.py:147-151), so this is how `Service.search()` uses a whole node."""

    supports_fused_normalize = True  # search(..., normalize=True) fuses faiss.normalize_L2

    def __init__(self, d: int, dtype: Any = "f32", device: int = 0, base: int = 0,
                 devices: Any = None, replicate: bool = False):
        if d <= 0:
            raise ValueError("d must be positive")
        self.d = int(d)
        self._dtype = _dtype_code(dtype)
        self.devices = _device_list(devices)
        # replicate=True: every device of `devices` holds the WHOLE corpus and synchronous searches
        # are dealt round-robin to the replicas (ls_create_replicated) instead of row shards
        self.replicate = bool(replicate)
        if self.replicate and not self.devices:
            raise ValueError("replicate=True needs devices=[...]")
        self.device = int(self.devices[0]) if self.devices else int(device)
        self._base = int(base)
        self._handle: ctypes.c_void_p | None = None
        # rows added before the first search: uploaded (and released) when the handle is built.
        # No host copy outlives that: later add() calls append in HBM (ls_add), host_corpus()
        # reads the rows back (ls_reconstruct).
        self._pending: list[np.ndarray] = []
        self._ntotal = 0
        self._device_built = False             # built straight from device memory
        self.is_trained = True

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_array(cls, corpus: np.ndarray, dtype: Any = "f32", device: int = 0,
                   base: int = 0, devices: Any = None, replicate: bool = False) -> "FlatIPIndex":
        corpus = np.asarray(corpus)
        if corpus.ndim != 2:
            raise ValueError("corpus must be [n, d]")
        ix = cls(corpus.shape[1], dtype=dtype, device=device, base=base, devices=devices,
                 replicate=replicate)
        ix.add(corpus)
        ix._ensure_built()
        return ix

    @classmethod
    def from_device_tensor(cls, corpus, dtype: Any = "f32", base: int = 0) -> "FlatIPIndex":
        """Build from a torch float32 CUDA tensor [n, d] without a host round trip."""
        import torch

        if not (isinstance(corpus, torch.Tensor) and corpus.is_cuda and corpus.dim() == 2
                and corpus.dtype == torch.float32 and corpus.is_contiguous()):
            raise ValueError("expected a contiguous float32 CUDA tensor [n, d]")
        ix = cls(corpus.shape[1], dtype=dtype, device=corpus.device.index or 0, base=base)
        lib = native.load()
        h = ctypes.c_void_p()
        torch.cuda.synchronize(corpus.device)
        native.check(lib.ls_create_from_device(ctypes.byref(h), corpus.data_ptr(),
                                               corpus.shape[0], ix.d, ix._dtype, ix.device))
        ix._handle = h
        ix._ntotal = int(corpus.shape[0])
        ix._device_built = True
        if base:
            native.check(lib.ls_set_base(h, base))
        return ix

    @classmethod
    def from_device_blocks(cls, blocks: list, dtype: Any = "f32", base: int = 0) -> "FlatIPIndex":
        """Row-sharded index from per-device float32 CUDA tensors ``blocks[g]`` of shape
        [rows_g, d] (block g stays on its own GPU; global rows are numbered block after block).
        Multi-GB synthetic shards never cross the host (ls_create_sharded_from_device)."""
        import torch

        if not blocks:
            raise ValueError("need at least one block")
        d = int(blocks[0].shape[1])
        for b in blocks:
            if not (isinstance(b, torch.Tensor) and b.is_cuda and b.dim() == 2 and b.shape[1] == d
                    and b.dtype == torch.float32 and b.is_contiguous()):
                raise ValueError("every block must be a contiguous float32 CUDA tensor [rows, d]")
        devs = [b.device.index or 0 for b in blocks]
        ix = cls(d, dtype=dtype, base=base, devices=devs)
        n = len(blocks)
        ptrs = (ctypes.c_void_p * n)(*[b.data_ptr() for b in blocks])
        rows = (ctypes.c_int64 * n)(*[int(b.shape[0]) for b in blocks])
        ids = (ctypes.c_int32 * n)(*devs)
        for b in blocks:
            torch.cuda.synchronize(b.device)
        h = ctypes.c_void_p()
        lib = native.load()
        native.check(lib.ls_create_sharded_from_device(ctypes.byref(h), ptrs, rows, d, ix._dtype, ids, n))
        ix._handle = h
        ix._ntotal = int(sum(int(b.shape[0]) for b in blocks))
        ix._device_built = True
        if base:
            native.check(lib.ls_set_base(h, base))
        return ix

    def shards(self) -> list[tuple[int, int, int]]:
        """(device, first global row, rows) of every shard; [] for a single-device index."""
        h = self._ensure_built()
        lib = native.load()
        out = []
        for g in range(int(lib.ls_shard_count(h))):
            dev, row0, rows = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int64()
            native.check(lib.ls_shard_info(h, g, ctypes.byref(dev), ctypes.byref(row0),
                                           ctypes.byref(rows)))
            out.append((dev.value, row0.value, rows.value))
        return out

    def exchange_info(self) -> dict:
        """What the exchange step of a sharded handle does on this node (ls_shard_exchange_info):
        transport, RCCL version / failure text, enqueue workers, peer-access matrix. {} for a
        single-device index."""
        import json

        h = self._ensure_built()
        lib = native.load()
        if int(lib.ls_shard_count(h)) == 0:
            return {}
        buf = ctypes.create_string_buffer(8192)
        need = int(lib.ls_shard_exchange_info(h, buf, len(buf)))
        if need < 0:
            native.check(need)
        return json.loads(buf.value.decode())

    def add(self, x: np.ndarray) -> None:
        """index.add(x) (reference extract/index.py:116): append float32 rows. On a built index
        the stored rows stay in HBM and only the new ones are uploaded (ls_add)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError(f"add expects [n, {self.d}] float32")
        if x.shape[0] == 0:
            return
        if self._handle is not None:
            native.check(native.load().ls_add(self._handle, x.ctypes.data, x.shape[0]))
        else:
            self._pending.append(x)
        self._ntotal += x.shape[0]

    def _drop_handle(self) -> None:
        if self._handle is not None:
            native.load().ls_destroy(self._handle)
            self._handle = None

    def _ensure_built(self) -> ctypes.c_void_p:
        if self._handle is not None:
            return self._handle
        lib = native.load()
        if len(self._pending) > 1:
            self._pending = [np.concatenate(self._pending, axis=0)]
        corpus = self._pending[0] if self._pending else np.zeros((0, self.d), np.float32)
        h = ctypes.c_void_p()
        if self.devices is not None:
            ids = (ctypes.c_int32 * len(self.devices))(*self.devices)
            create = lib.ls_create_replicated if self.replicate else lib.ls_create_sharded
            native.check(create(
                ctypes.byref(h), corpus.ctypes.data if corpus.size else None, corpus.shape[0],
                self.d, self._dtype, ids, len(self.devices)))
        else:
            native.check(lib.ls_create(ctypes.byref(h), corpus.ctypes.data if corpus.size else None,
                                       corpus.shape[0], self.d, self._dtype, self.device))
        self._handle = h
        self._pending = []  # the rows live in HBM now
        if self._base:
            native.check(lib.ls_set_base(h, self._base))
        return h

    # ------------------------------------------------------------------ properties
    @property
    def ntotal(self) -> int:
        return self._ntotal

    @property
    def storage_dtype(self) -> str:
        return "f16" if self._dtype == native.LS_DTYPE_F16 else "f32"

    @property
    def base(self) -> int:
        return self._base

    def host_corpus(self) -> np.ndarray:
        """The stored rows as float32 [ntotal, d], read back from HBM (index.reconstruct_n);
        an fp16 index returns the rounded values. Before the first search the rows have not been
        uploaded yet and are returned as added."""
        if self._handle is None and not self._device_built:
            if len(self._pending) > 1:
                self._pending = [np.concatenate(self._pending, axis=0)]
            return self._pending[0] if self._pending else np.zeros((0, self.d), np.float32)
        h = self._ensure_built()
        out = np.empty((self._ntotal, self.d), dtype=np.float32)
        if self._ntotal:
            native.check(native.load().ls_reconstruct(h, 0, self._ntotal, out.ctypes.data))
        return out

    # ------------------------------------------------------------------ search
    def search(self, x: np.ndarray, k: int, *, normalize: bool = False
               ) -> tuple[np.ndarray, np.ndarray]:
        """index.search(x, k) (reference search/engine.py:250).

        x: float32 [nq, d]. Returns (D float32 [nq, k], I int64 [nq, k]) best first under
        (score desc, row asc); unfilled slots are (-FLT_MAX, -1).
        """
        # (this wrapper is ~2 us of a 62 us call: no conversion call for the usual float32 row-major
        # query, no helper objects for the three pointers - native.addr)
        if type(x) is not np.ndarray or x.dtype != np.float32 or not x.flags.c_contiguous:
            x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError(f"search expects [nq, {self.d}] float32, got {x.shape}")
        k = int(k)
        if k <= 0:
            raise ValueError("k must be positive")
        h = self._handle if self._handle is not None else self._ensure_built()
        nq = x.shape[0]
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        flags = native.LS_FLAG_NORMALIZE if normalize else 0
        fast = native.fast_search()
        if fast is not None:  # csrc/lsfast.c: the same ls_search, bound without ctypes' argument conversion
            rc = fast[0].search(fast[1], h.value, x, nq, k, flags, D, I)
        else:
            addr = native.addr
            rc = native.load().ls_search(h, addr(x), nq, k, flags, addr(D), addr(I))
        if rc:
            native.check(rc)
        return D, I

    def search_device(self, q, k: int, out_scores=None, out_indices=None, *,
                      normalize: bool = False, asynchronous: bool = False,
                      pipeline: bool = False, inorder: bool = False, stream=None):
        """Search with torch CUDA tensors (queries and results stay in HBM).

        q: float32 CUDA tensor [nq, d]. Work is queued on ``stream`` (default: torch's current
        stream). Returns (scores float32 [nq, k], indices int64 [nq, k]) CUDA tensors.
        ``asynchronous``: queue and return, results ordered on the stream. For ANY batched call
        (the speculative MFMA paths: nq > 16 on an fp16 index, nq > 32 on an fp32 index) call
        :meth:`check` before trusting them: it repairs the rare query the speculative threshold
        short-changed, re-writing its rows of the output tensors (keep those alive until then;
        ``q`` may be reused at once in stream order). On a sharded index (``devices=[...]``) ``q``
        and the outputs live on ``devices[0]``.
        ``pipeline``: queue on the index's internal lanes so consecutive calls overlap; results
        are valid only after :meth:`check` (scan-path launches write no score vectors: a query whose
        selection could not prove its result complete is served again in place there). ``inorder``
        (LS_FLAG_INORDER): the caller consumes pipelined scan-path results on the GPU before
        :meth:`check`; launches then keep their score vectors and are exact in the lanes' order.
        """
        import torch

        if not (q.is_cuda and q.dtype == torch.float32 and q.dim() == 2 and q.is_contiguous()
                and q.shape[1] == self.d):
            raise ValueError(f"search_device expects a contiguous float32 CUDA tensor [nq, {self.d}]")
        nq = q.shape[0]
        if out_scores is None:
            out_scores = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        if out_indices is None:
            out_indices = torch.empty((nq, k), dtype=torch.int64, device=q.device)
        h = self._ensure_built()
        s = stream if stream is not None else torch.cuda.current_stream(q.device)
        flags = (native.LS_FLAG_NORMALIZE if normalize else 0) | \
                (native.LS_FLAG_ASYNC if asynchronous else 0) | \
                (native.LS_FLAG_PIPELINE if pipeline else 0) | \
                (native.LS_FLAG_INORDER if inorder else 0)
        native.check(native.load().ls_search_device(h, q.data_ptr(), nq, int(k), flags,
                                                    out_scores.data_ptr(), out_indices.data_ptr(),
                                                    s.cuda_stream))
        return out_scores, out_indices

    def export_flags(self, dst, stream=None) -> None:
        """Copy the per-query verification flags (uint32/int32 CUDA tensor [nq]) of the most
        recent search queued on this index, in stream order (see ls_export_flags)."""
        import torch

        if not (dst.is_cuda and dst.dim() == 1 and dst.element_size() == 4 and dst.is_contiguous()):
            raise ValueError("export_flags expects a contiguous 32-bit CUDA tensor [nq]")
        s = stream if stream is not None else torch.cuda.current_stream(dst.device)
        native.check(native.load().ls_export_flags(self._ensure_built(), dst.data_ptr(),
                                                   dst.shape[0], s.cuda_stream))

    def check(self, stream=None) -> None:
        """Synchronise and validate async searches (see ls_check)."""
        import torch

        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        native.check(native.load().ls_check(self._ensure_built(), s.cuda_stream))

    # ------------------------------------------------------------------ instrumentation
    def set_profiling(self, enabled: bool) -> None:
        native.check(native.load().ls_set_profiling(self._ensure_built(), 1 if enabled else 0))

    def last_kernel_ms(self) -> tuple[float, float]:
        a, b = ctypes.c_float(), ctypes.c_float()
        native.check(native.load().ls_last_kernel_ms(self._ensure_built(), ctypes.byref(a),
                                                     ctypes.byref(b)))
        return a.value, b.value

    def debug_option(self, which: int, value: int) -> None:
        native.check(native.load().ls_debug_option(self._ensure_built(), which, value))

    def debug_counter(self, which: int) -> int:
        return int(native.load().ls_debug_counter(self._ensure_built(), which))

    def debug_scores(self) -> np.ndarray:
        """Score vector S of the most recent per-query scan (test hook)."""
        out = np.empty(self._ntotal, dtype=np.float32)
        native.check(native.load().ls_debug_read_scores(self._ensure_built(), out.ctypes.data,
                                                        out.size))
        return out

    def close(self) -> None:
        self._drop_handle()

    def __del__(self):
        try:
            self._drop_handle()
        except Exception:
            pass
