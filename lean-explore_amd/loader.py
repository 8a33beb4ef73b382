"""Corpus loaders: build the fp32 matrix C[N, d] and the row -> declaration-id map from the
reference's on-disk artefacts, without faiss or SQLAlchemy.

* SQLite route: `declarations.informalization_embedding` holds each vector as a native
  little-endian float32 blob (`struct.pack(f"{n}f", *values)`, reference
  src/lean_explore/models/search_db.py:24-35). Rows are taken exactly as the reference's index
  builder takes them (reference src/lean_explore/extract/index.py:59-71: id + embedding WHERE the
  embedding IS NOT NULL, table order, no ORDER BY), so row i pairs with ids[i] as in
  extract/index.py:176-181.
* ids JSON: `informalization_faiss_ids_map.json` is a JSON list of ints (extract/index.py:176-181).
"""

from __future__ import annotations

import json
import sqlite3
import struct
from pathlib import Path

import numpy as np

EMBEDDING_COLUMN = "informalization_embedding"
TABLE = "declarations"


def embedding_to_blob(values) -> bytes:
    """list[float] -> bytes, the BinaryEmbedding bind format (search_db.py:24-28)."""
    vals = list(values)
    return struct.pack(f"{len(vals)}f", *vals)


def blob_to_embedding(blob: bytes) -> np.ndarray:
    """bytes -> float32 vector, the BinaryEmbedding result format (search_db.py:30-35)."""
    return np.frombuffer(blob, dtype="<f4", count=len(blob) // 4).astype(np.float32, copy=True)


def load_corpus_from_sqlite(db_path: str | Path, column: str = EMBEDDING_COLUMN
                            ) -> tuple[list[int], np.ndarray]:
    """(declaration_ids, C[N, d] float32). Empty database -> ([], array of shape (0,)) like the
    reference (extract/index.py:63-65)."""
    if not column.replace("_", "").isalnum():
        raise ValueError("bad column name")
    con = sqlite3.connect(f"file:{Path(db_path)}?mode=ro", uri=True)
    try:
        rows = con.execute(
            f"SELECT id, {column} FROM {TABLE} WHERE {column} IS NOT NULL").fetchall()
    finally:
        con.close()
    if not rows:
        return [], np.array([])
    ids = [int(r[0]) for r in rows]
    d = len(rows[0][1]) // 4
    corpus = np.empty((len(rows), d), dtype=np.float32)
    for i, (_, blob) in enumerate(rows):
        if len(blob) != 4 * d:
            raise ValueError(f"row {ids[i]}: embedding has {len(blob) // 4} dims, expected {d}")
        corpus[i] = np.frombuffer(blob, dtype="<f4")
    return ids, corpus


def load_ids_map(path: str | Path) -> list[int]:
    with open(path) as f:
        ids = json.load(f)
    if not isinstance(ids, list):
        raise ValueError("ids map must be a JSON list")
    return [int(x) for x in ids]


def save_ids_map(path: str | Path, ids: list[int]) -> None:
    with open(path, "w") as f:
        json.dump([int(x) for x in ids], f)
