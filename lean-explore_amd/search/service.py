"""Service — the backend object the MCP tools and the CLI talk to; same signatures and defaults
as the reference (reference src/lean_explore/search/service.py:15-68: limit=20, rerank_top=50)."""

from __future__ import annotations

import time

from .engine import SearchEngine
from .types import SearchResponse, SearchResult


class Service:
    def __init__(self, engine: SearchEngine | None = None):
        # reference service.py:21: no engine -> the default one (cache paths from the environment)
        self.engine = engine or SearchEngine()

    async def search(self, query: str, limit: int = 20, rerank_top: int | None = 50,
                     packages: list[str] | None = None) -> SearchResponse:
        start = time.time()
        results = await self.engine.search(query=query, limit=limit, rerank_top=rerank_top,
                                           packages=packages)
        return SearchResponse(query=query, results=results, count=len(results),
                              processing_time_ms=int((time.time() - start) * 1000))

    async def get_by_id(self, declaration_id: int) -> SearchResult | None:
        return await self.engine.get_by_id(declaration_id)
