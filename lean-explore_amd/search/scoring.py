"""Score fusion helpers used after the dense retrieval — behaviour of the reference's
src/lean_explore/search/scoring.py:14-156, pinned by tests/golden/reference_vectors.json.
Plain Python on <= 1000 items; not a kernel."""

from __future__ import annotations

import math
from difflib import SequenceMatcher

EPSILON = 1e-9


def normalize_scores(scores: list[float]) -> list[float]:
    """Min-max to [0, 1]; a flat list maps to all 1.0 if positive else all 0.0 (scoring.py:14-37)."""
    if not scores:
        return []
    lo, hi = min(scores), max(scores)
    span = hi - lo
    if span < EPSILON:
        return [1.0 if hi > EPSILON else 0.0] * len(scores)
    return [(s - lo) / span for s in scores]


def normalize_dependency_counts(counts: list[int]) -> list[float]:
    """log(1+c) / log(1+max) (scoring.py:40-60)."""
    if not counts:
        return []
    top = max(counts)
    if top == 0:
        return [0.0] * len(counts)
    denom = math.log(1 + top)
    return [math.log(1 + c) / denom for c in counts]


def compute_ranks(scores: list[float]) -> list[int]:
    """1 = best; non-positive scores get rank len+1 (scoring.py:63-83). Stable for ties."""
    n = len(scores)
    order = sorted(range(n), key=lambda i: scores[i], reverse=True)
    ranks = [0] * n
    for position, i in enumerate(order, start=1):
        ranks[i] = position if scores[i] > 0 else n + 1
    return ranks


def reciprocal_rank_fusion(rank_lists: list[list[int]], k: int = 0) -> list[float]:
    """sum_i 1 / (k + rank_i) (scoring.py:86-104)."""
    return [sum(1.0 / (k + ranks[i]) for ranks in rank_lists) for i in range(len(rank_lists[0]))]


def weighted_score_fusion(score_lists: list[list[float]], weights: list[float]) -> list[float]:
    """Weighted sum of min-max normalised signals (scoring.py:107-138)."""
    if not score_lists or not score_lists[0]:
        return []
    normed = [normalize_scores(s) for s in score_lists]
    return [sum(w * normed[j][i] for j, w in enumerate(weights)) for i in range(len(score_lists[0]))]


def fuzzy_name_score(query: str, name: str) -> float:
    """difflib ratio after mapping dots/underscores to spaces (scoring.py:141-156)."""
    def flat(s: str) -> str:
        return s.lower().replace(".", " ").replace("_", " ")

    return SequenceMatcher(None, flat(query), flat(name)).ratio()
