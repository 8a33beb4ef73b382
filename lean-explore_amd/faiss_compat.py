"""The slice of the `faiss` module surface that LeanExplore's local backend uses
(reference src/lean_explore/search/engine.py:156-159,240-250), served by the HIP library:

    import lean_explore_amd.faiss_compat as faiss
    index = faiss.read_index(path); faiss.normalize_L2(x); D, I = index.search(x, k)

Index files. `write_index` / `read_index` speak FAISS's own container for flat inner-product
indexes (fourcc ``IxFI``) and `read_index` also accepts the IVF-flat container the reference ships
(fourcc ``IwFl``, reference src/lean_explore/extract/index.py:103-104,173): the inverted lists are
scattered back to add order and searched exactly (what IVF approximates). The byte layout is
restated from upstream faiss's index_write.cpp; no faiss-written file exists in this image, so it
is checked by writer/reader round trips only — UNVERIFIED against real faiss files.
"""

from __future__ import annotations

import struct
from pathlib import Path

import numpy as np

from .index import FlatIPIndex, normalize_L2  # noqa: F401  (re-exported)

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1


class IndexFlatIP(FlatIPIndex):
    """faiss.IndexFlatIP(d) (reference extract/index.py:103)."""

    def __init__(self, d: int, dtype="f32", device: int = 0):
        super().__init__(d, dtype=dtype, device=device)


class IndexIVFFlat(FlatIPIndex):
    """faiss.IndexIVFFlat(quantizer, d, nlist, metric) as the reference's index builder drives it
    (reference src/lean_explore/extract/index.py:103-116: construct, ``train``, ``add``;
    ``nprobe`` is set by the engine, search/engine.py:247-248). There is nothing to train: the
    rows are searched exactly, i.e. the answer IVF approximates; ``nlist`` / ``nprobe`` are kept
    as plain attributes. ``write_index`` stores it in the flat container."""

    def __init__(self, quantizer, d: int, nlist: int, metric: int = METRIC_INNER_PRODUCT,
                 dtype="f32", device: int = 0):
        if metric != METRIC_INNER_PRODUCT:
            raise ValueError("only METRIC_INNER_PRODUCT is supported")
        super().__init__(d, dtype=dtype, device=device)
        self.quantizer, self.nlist, self.nprobe, self.is_trained = quantizer, int(nlist), 1, False

    def train(self, x: np.ndarray) -> None:
        self.is_trained = True


def get_num_gpus() -> int:
    """faiss.get_num_gpus(): 0, so callers take their CPU-index code path (extract/index.py:106);
    the rows go to HBM when the index is first searched either way."""
    return 0


# ------------------------------------------------------------------ container format helpers
def _fourcc(s: str) -> int:
    return struct.unpack("<I", s.encode("ascii"))[0]


def _write_header(f, d: int, ntotal: int, metric: int) -> None:
    # write_index_header: d (int32), ntotal (int64), two dummy int64 (1 << 20), is_trained (u8),
    # metric_type (int32)
    f.write(struct.pack("<iqqqBi", d, ntotal, 1 << 20, 1 << 20, 1, metric))


def _read_header(f) -> tuple[int, int, int]:
    d, ntotal, _, _, _, metric = struct.unpack("<iqqqBi", f.read(4 + 8 + 8 + 8 + 1 + 4))
    if metric > 1:
        f.read(4)  # metric_arg
    return d, ntotal, metric


def _read_vector(f, dtype, what: str) -> np.ndarray:
    (count,) = struct.unpack("<Q", f.read(8))
    itemsize = np.dtype(dtype).itemsize
    raw = f.read(count * itemsize)
    if len(raw) != count * itemsize:
        raise ValueError(f"truncated index file while reading {what}")
    return np.frombuffer(raw, dtype=dtype, count=count)


def _read_flat_payload(f) -> tuple[int, np.ndarray]:
    d, ntotal, _ = _read_header(f)
    xb = _read_vector(f, "<f4", "flat storage")  # count is in floats (xb / codes/4)
    if xb.size != d * ntotal:
        raise ValueError("flat index: storage size does not match d * ntotal")
    return d, xb.reshape(ntotal, d).astype(np.float32, copy=True)


def write_index(index: FlatIPIndex, path: str | Path, *, allow_lossy: bool = False) -> None:
    """faiss.write_index for a flat inner-product index (container ``IxFI``).

    A built fp16 index holds only the ROUNDED rows in HBM (no host copy is kept), so the file
    would not round-trip the float32 embeddings that were added and would score differently in
    the reference's faiss: refused unless ``allow_lossy=True``."""
    if (getattr(index, "storage_dtype", "f32") == "f16" and getattr(index, "_handle", None) is not None
            and not allow_lossy):
        raise ValueError("write_index on a built fp16 index would write fp16-rounded rows, not the "
                         "float32 embeddings that were added; pass allow_lossy=True to do that, or "
                         "write the index before the first search / from an f32 index")
    corpus = np.ascontiguousarray(index.host_corpus(), dtype="<f4")
    with open(path, "wb") as f:
        f.write(struct.pack("<I", _fourcc("IxFI")))
        _write_header(f, index.d, corpus.shape[0], METRIC_INNER_PRODUCT)
        f.write(struct.pack("<Q", corpus.size))
        f.write(corpus.tobytes())


def read_index(path: str | Path, dtype="f32", device: int = 0, devices=None,
               replicate: bool = False) -> FlatIPIndex:
    """faiss.read_index (reference search/engine.py:159) -> exact HIP index (``devices``: row-sharded
    over several GPUs inside this process)."""
    path = Path(path)
    with open(path, "rb") as f:
        (cc,) = struct.unpack("<I", f.read(4))
        if cc == _fourcc("IxFI"):
            d, corpus = _read_flat_payload(f)
        elif cc == _fourcc("IwFl"):
            d, corpus = _read_ivf_flat(f)
        else:
            tag = struct.pack("<I", cc).decode("ascii", "replace")
            raise ValueError(f"{path}: unsupported index container {tag!r} "
                             "(expected IxFI flat-IP or IwFl IVF-flat)")
    index = FlatIPIndex(d, dtype=dtype, device=device, devices=devices, replicate=replicate)
    if corpus.shape[0]:
        index.add(corpus)
    return index


def _read_ivf_flat(f) -> tuple[int, np.ndarray]:
    """IndexIVFFlat: ivf header, nested quantizer, direct map, ArrayInvertedLists (``ilar``)."""
    d, ntotal, metric = _read_header(f)
    if metric != METRIC_INNER_PRODUCT:
        raise ValueError("IVF index is not an inner-product index")
    nlist, _nprobe = struct.unpack("<QQ", f.read(16))
    (qcc,) = struct.unpack("<I", f.read(4))  # nested coarse quantizer: read and discard
    if qcc not in (_fourcc("IxFI"), _fourcc("IxF2")):
        raise ValueError("IVF index: unsupported coarse quantizer container")
    _read_flat_payload(f)
    (dm_type,) = struct.unpack("<b", f.read(1))  # direct map
    _read_vector(f, "<i8", "direct map")
    if dm_type == 2:
        raise ValueError("IVF index: hashtable direct maps are not supported")
    (lcc,) = struct.unpack("<I", f.read(4))
    if lcc != _fourcc("ilar"):
        raise ValueError("IVF index: unsupported inverted-list container")
    nl, code_size = struct.unpack("<QQ", f.read(16))
    if nl != nlist or code_size != 4 * d:
        raise ValueError("IVF index: inverted lists do not match the header")
    (list_type,) = struct.unpack("<I", f.read(4))
    sizes = np.zeros(nlist, dtype=np.int64)
    raw = _read_vector(f, "<u8", "list sizes")
    if list_type == _fourcc("full"):
        sizes[:] = raw
    elif list_type == _fourcc("sprs"):
        sizes[raw[0::2].astype(np.int64)] = raw[1::2]
    else:
        raise ValueError("IVF index: unknown list-size encoding")
    corpus = np.zeros((ntotal, d), dtype=np.float32)
    seen = np.zeros(ntotal, dtype=bool)
    for n in sizes:
        n = int(n)
        if n == 0:
            continue
        codes = np.frombuffer(f.read(n * code_size), dtype="<f4").reshape(n, d)
        ids = np.frombuffer(f.read(n * 8), dtype="<i8")
        corpus[ids] = codes  # ids are the add-order row numbers (index.add without ids)
        seen[ids] = True
    if not seen.all():
        raise ValueError("IVF index: inverted lists do not cover every row")
    return d, corpus
