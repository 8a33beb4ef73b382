"""Where the local backend finds its data: the same environment variables, directory layout and
file names as the reference (reference src/lean_explore/config.py:11-30 active version,
:33-41 data directory, :84-106 active data path, :109-196 `Config`), so a cache populated by the
reference's `lean-explore data fetch` is found without configuration.

Resolution happens when `resolve()` is called (the reference freezes the same values at import
time); everything else is identical:

* cache:  ``$LEAN_EXPLORE_CACHE_DIR`` or ``~/.lean_explore/cache``, then ``/<active version>``
* version: ``$LEAN_EXPLORE_VERSION``, else ``~/.lean_explore/active_version``, else ``v4.24.0``
* data:   ``$LEAN_EXPLORE_DATA_DIR`` or ``<repo root>/data``; the active data path is that directory
  itself if it holds ``lean_explore.db``, else the newest complete ``YYYYMMDD_HHMMSS`` extraction
  directory, else ``<data>/<active version>``
"""

from __future__ import annotations

import os
import re
from dataclasses import dataclass
from pathlib import Path

DEFAULT_VERSION = "v4.24.0"
DB_FILE = "lean_explore.db"
FAISS_INDEX_FILE = "informalization_faiss.index"
FAISS_IDS_MAP_FILE = "informalization_faiss_ids_map.json"
BM25_SPACED_DIR = "bm25_name_spaced"
BM25_RAW_DIR = "bm25_name_raw"
BM25_IDS_MAP_FILE = "bm25_ids_map.json"
# what makes an extraction directory complete (reference config.py:61-68)
REQUIRED_FILES = (DB_FILE, FAISS_INDEX_FILE, FAISS_IDS_MAP_FILE, BM25_IDS_MAP_FILE, BM25_RAW_DIR,
                  BM25_SPACED_DIR)
_STAMP = re.compile(r"^\d{8}_\d{6}$")


def active_version() -> str:
    env = os.getenv("LEAN_EXPLORE_VERSION")
    if env:
        return env
    marker = Path.home() / ".lean_explore" / "active_version"
    return marker.read_text().strip() if marker.exists() else DEFAULT_VERSION


def cache_directory() -> Path:
    return Path(os.getenv("LEAN_EXPLORE_CACHE_DIR", Path.home() / ".lean_explore" / "cache"))


def data_directory() -> Path:
    return Path(os.getenv("LEAN_EXPLORE_DATA_DIR", Path(__file__).resolve().parent.parent / "data"))


def active_data_path(data_dir: Path, version: str) -> Path:
    if (data_dir / DB_FILE).exists():
        return data_dir
    if data_dir.exists():
        stamped = sorted((p for p in data_dir.iterdir() if p.is_dir() and _STAMP.match(p.name)),
                         key=lambda p: p.name, reverse=True)
        for cand in stamped:
            if all((cand / name).exists() for name in REQUIRED_FILES):
                return cand
    return data_dir / version


@dataclass(frozen=True)
class Paths:
    base_path: Path
    database_path: Path
    database_url: str


def resolve(use_local_data: bool = False) -> Paths:
    """``use_local_data=False``: the downloaded cache (reference `Config.ACTIVE_CACHE_PATH`,
    `DATABASE_URL`); ``True``: locally extracted data (`ACTIVE_DATA_PATH`,
    `EXTRACTION_DATABASE_URL`). Reference search/engine.py:82-87."""
    version = active_version()
    base = active_data_path(data_directory(), version) if use_local_data \
        else cache_directory() / version
    db = base / DB_FILE
    return Paths(base_path=base, database_path=db, database_url=f"sqlite+aiosqlite:///{db}")


def sqlite_path_from_url(db_url: str) -> Path:
    """The file behind the SQLAlchemy URLs the reference passes around
    (``sqlite+aiosqlite:///<path>``, reference config.py:193). Only SQLite is supported: the
    local backend's storage is a single file read with the standard library."""
    m = re.match(r"^sqlite(\+\w+)?:///(.*)$", str(db_url))
    if not m:
        if "://" in str(db_url):
            raise ValueError(f"unsupported database URL {db_url!r}: the local backend reads SQLite "
                             "files (sqlite+aiosqlite:///<path>)")
        return Path(db_url)
    return Path(m.group(2))


class _ConfigMeta(type):
    """Attribute names of the reference's `Config` (reference config.py:109-196), resolved at access
    time from the same environment variables: code written against `Config.ACTIVE_CACHE_PATH`,
    `Config.DATABASE_URL`, ... keeps working."""

    CACHE_DIRECTORY = property(lambda cls: cache_directory())
    DATA_DIRECTORY = property(lambda cls: data_directory())
    ACTIVE_VERSION = property(lambda cls: active_version())
    ACTIVE_CACHE_PATH = property(lambda cls: resolve(False).base_path)
    ACTIVE_DATA_PATH = property(lambda cls: resolve(True).base_path)
    DATABASE_PATH = property(lambda cls: resolve(False).database_path)
    DATABASE_URL = property(lambda cls: resolve(False).database_url)
    EXTRACTION_DATABASE_PATH = property(lambda cls: resolve(True).database_path)
    EXTRACTION_DATABASE_URL = property(lambda cls: resolve(True).database_url)
    FAISS_INDEX_PATH = property(lambda cls: resolve(False).base_path / FAISS_INDEX_FILE)
    FAISS_IDS_MAP_PATH = property(lambda cls: resolve(False).base_path / FAISS_IDS_MAP_FILE)
    BM25_SPACED_PATH = property(lambda cls: resolve(False).base_path / BM25_SPACED_DIR)
    BM25_RAW_PATH = property(lambda cls: resolve(False).base_path / BM25_RAW_DIR)
    BM25_IDS_MAP_PATH = property(lambda cls: resolve(False).base_path / BM25_IDS_MAP_FILE)


class Config(metaclass=_ConfigMeta):
    DEFAULT_LEAN_VERSION = "4.24.0"  # reference config.py:131
