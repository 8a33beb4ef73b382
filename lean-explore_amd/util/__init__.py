"""Model clients that sit either side of the dense lookup. They stay on
PyTorch-ROCm by design (BASELINE north_star): no HIP kernels here, only the reference's call
surface so that `SearchEngine` can be driven end to end on the GPU box."""

from .embedding_client import EmbeddingClient, EmbeddingResponse
from .reranker_client import RerankerClient, RerankerResponse

__all__ = ["EmbeddingClient", "EmbeddingResponse", "RerankerClient", "RerankerResponse"]
