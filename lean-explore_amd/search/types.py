"""Result types of the search surface — field-for-field the reference's pydantic models
(reference src/lean_explore/models/search_types.py:61-108), so `Service.search()` callers (MCP
tools, CLI) see no difference."""

from __future__ import annotations

from pydantic import BaseModel, ConfigDict


class SearchResult(BaseModel):
    """One Lean declaration returned by a search (search_types.py:61-92)."""

    id: int
    name: str
    module: str
    docstring: str | None
    source_text: str
    source_link: str
    dependencies: str | None
    informalization: str | None

    model_config = ConfigDict(from_attributes=True)


class SearchResponse(BaseModel):
    """Results plus metadata (search_types.py:95-108)."""

    query: str
    results: list[SearchResult]
    count: int
    processing_time_ms: int | None = None
