"""ctypes binding of libleansearch.so (include/leansearch.h) — the only way Python reaches the
HIP kernels. There is no fallback: if the shared library is missing, or no MI355X is visible,
every compute call raises.

The reference reaches its dense index through the faiss SWIG module
(reference src/lean_explore/search/engine.py:156,240); this module is the counterpart of that
import for the HIP library.
"""

from __future__ import annotations

import ctypes
import os
from pathlib import Path

LS_OK = 0
LS_ERR_INVALID_ARG = -1
LS_ERR_NO_DEVICE = -2
LS_ERR_HIP = -3
LS_ERR_K_TOO_LARGE = -4
LS_ERR_OVERFLOW = -5

LS_DTYPE_F32 = 0
LS_DTYPE_F16 = 1
LS_FLAG_NORMALIZE = 1
LS_FLAG_ASYNC = 2
LS_FLAG_PIPELINE = 4
LS_FLAG_INORDER = 8
LS_MAX_K = 2048

_PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("LEANSEARCH_LIB", _PKG_DIR / "libleansearch.so"))

# every symbol include/leansearch.h declares: (restype, argtypes)
_vp = ctypes.c_void_p
_i32 = ctypes.c_int32
_i64 = ctypes.c_int64
_u32 = ctypes.c_uint32
_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)
SYMBOLS: dict[str, tuple] = {
    "ls_create": (ctypes.c_int, [ctypes.POINTER(_vp), _vp, _i64, _i32, _i32, _i32]),
    "ls_create_from_device": (ctypes.c_int, [ctypes.POINTER(_vp), _vp, _i64, _i32, _i32, _i32]),
    "ls_create_sharded": (ctypes.c_int, [ctypes.POINTER(_vp), _vp, _i64, _i32, _i32, _vp, _i32]),
    "ls_create_replicated": (ctypes.c_int, [ctypes.POINTER(_vp), _vp, _i64, _i32, _i32, _vp, _i32]),
    "ls_create_sharded_from_device": (ctypes.c_int, [ctypes.POINTER(_vp), _vp, _vp, _i32, _i32, _vp,
                                                     _i32]),
    "ls_shard_count": (_i32, [_vp]),
    "ls_shard_info": (ctypes.c_int, [_vp, _i32, ctypes.POINTER(_i32), _i64p, _i64p]),
    "ls_shard_exchange_info": (_i32, [_vp, ctypes.c_char_p, _i32]),
    "ls_add": (ctypes.c_int, [_vp, _vp, _i64]),
    "ls_reconstruct": (ctypes.c_int, [_vp, _i64, _i64, _vp]),
    "ls_destroy": (None, [_vp]),
    "ls_ntotal": (_i64, [_vp]),
    "ls_dim": (_i32, [_vp]),
    "ls_dtype": (_i32, [_vp]),
    "ls_device": (_i32, [_vp]),
    "ls_set_base": (ctypes.c_int, [_vp, _i64]),
    "ls_search": (ctypes.c_int, [_vp, _vp, _i64, _i32, _u32, _vp, _vp]),
    "ls_search_device": (ctypes.c_int, [_vp, _vp, _i64, _i32, _u32, _vp, _vp, _vp]),
    "ls_check": (ctypes.c_int, [_vp, _vp]),
    "ls_export_flags": (ctypes.c_int, [_vp, _vp, _i64, _vp]),
    "ls_normalize_l2": (ctypes.c_int, [_vp, _i64, _i32, _i32]),
    "ls_merge_topk": (ctypes.c_int, [_vp, _vp, _i32, _i64, _i32, _vp, _vp, _i32, _vp]),
    "ls_merge_topk_strided": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i64, _i32, _vp, _vp, _i32, _vp]),
    "ls_set_profiling": (ctypes.c_int, [_vp, _i32]),
    "ls_last_kernel_ms": (ctypes.c_int, [_vp, _f32p, _f32p]),
    "ls_debug_option": (ctypes.c_int, [_vp, _i32, _i32]),
    "ls_debug_counter": (_i64, [_vp, _i32]),
    "ls_debug_read_scores": (ctypes.c_int, [_vp, _vp, _i64]),
    "ls_bm25_create": (ctypes.c_int, [ctypes.POINTER(_vp), _vp, _vp, _vp, _vp, _i64, _i64, _i32]),
    "ls_bm25_search": (ctypes.c_int, [_vp, _vp, _i32, _i32, _vp, _vp]),
    "ls_bm25_ntotal": (_i64, [_vp]),
    "ls_bm25_debug_counter": (_i64, [_vp, _i32]),
    "ls_bm25_destroy": (None, [_vp]),
    "ls_last_error": (ctypes.c_char_p, []),
    "ls_version": (ctypes.c_char_p, []),
    "ls_device_count": (_i32, []),
}

_lib: ctypes.CDLL | None = None


class LeanSearchError(RuntimeError):
    """A libleansearch call failed (carries the library's thread-local message)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libleansearch error {code}: {message}")
        self.code = code


def _preload_hip_runtime() -> None:
    """One HIP runtime per process. PyTorch wheels bundle their own libamdhip64.so.7 (same
    SONAME as /opt/rocm's); whichever is mapped first serves both, and torch cannot initialise
    on top of the system copy. So when torch is installed, map ITS runtime before ours resolves
    its NEEDED entry. Without torch, libleansearch binds to /opt/rocm as linked."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = Path(list(spec.submodule_search_locations)[0]) / "lib" / "libamdhip64.so"
    if cand.exists():
        ctypes.CDLL(str(cand), mode=ctypes.RTLD_GLOBAL)


def load() -> ctypes.CDLL:
    """Load libleansearch.so and bind every declared symbol. Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc, gfx950). There is no CPU fallback for the dense search path.")
    _preload_hip_runtime()
    lib = ctypes.CDLL(str(LIB_PATH))
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


_fast = None
_fast_tried = False


def fast_search():
    """(module, address of ls_search) of the CPython binding csrc/lsfast.c, or None: the same symbol
    of the same library as the ctypes binding, minus ~1.5 us of argument conversion per call."""
    global _fast, _fast_tried
    if not _fast_tried:
        _fast_tried = True
        try:
            from . import _lsfast
            _fast = (_lsfast, ctypes.cast(load().ls_search, ctypes.c_void_p).value)
        except ImportError:
            _fast = None
    return _fast


_c_char_from_buffer = ctypes.c_char.from_buffer
_addressof = ctypes.addressof


def addr(a) -> int | None:
    """Base address of a C-contiguous numpy array for a pointer argument (None if it is empty).
    ``ndarray.ctypes.data`` builds a helper object per access (1.4 us; three of them were most of
    the interpreter's share of a 64 us ``ls_search``); the buffer protocol gives the same address
    in 0.4 us. Read-only arrays do not export a writable buffer and take the slow way."""
    if not a.size:
        return None
    try:
        return _addressof(_c_char_from_buffer(a))
    except (TypeError, ValueError):
        return a.ctypes.data


def check(rc: int) -> None:
    if rc == LS_OK:
        return
    msg = load().ls_last_error().decode("utf-8", "replace")
    if rc == LS_ERR_INVALID_ARG:
        raise ValueError(f"libleansearch: {msg}")
    raise LeanSearchError(rc, msg)


def device_count() -> int:
    return int(load().ls_device_count())


def version() -> str:
    return load().ls_version().decode()
