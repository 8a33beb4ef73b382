"""Query/document embedding on PyTorch-ROCm.

Call surface of the reference's client (reference src/lean_explore/util/embedding_client.py:
`EmbeddingClient(model_name, device, max_length, batch_size)`, `async embed(texts, is_query)`
-> `EmbeddingResponse(texts, embeddings, model)`, :29-113). The reference delegates to
sentence-transformers, which is not installed in this image; Qwen3-Embedding's
sentence-transformers pipeline is Transformer -> last-token pooling -> L2 normalise with a
"query" prompt prefix, and that is what `encode` does directly on a `transformers` model.

`model` / `tokenizer` can be injected (tests and tools/hybrid_bench.py use a random-initialised
Qwen3 of the production shape and a synthetic tokenizer: no weights or vocab files can be
fetched here). Without them the named checkpoint is loaded with `from_pretrained`.
"""

from __future__ import annotations

import asyncio
import logging
import os
from typing import Any

import numpy as np
from pydantic import BaseModel

logger = logging.getLogger(__name__)

DEFAULT_BATCH_SIZE = 8  # reference embedding_client.py:13
# prompts["query"] of the Qwen3-Embedding sentence-transformers config
QUERY_PROMPT = ("Instruct: Given a web search query, retrieve relevant passages that answer "
                "the query\nQuery:")


class EmbeddingResponse(BaseModel):
    """reference embedding_client.py:16-26"""

    texts: list[str]
    embeddings: list[list[float]]
    model: str


class EmbeddingClient:
    def __init__(self, model_name: str, device: str | None = None, max_length: int | None = None,
                 batch_size: int | None = None, *, model: Any = None, tokenizer: Any = None,
                 dtype: Any = None, query_prompt: str = QUERY_PROMPT, use_graphs: bool = False, fused_norms: bool = False):
        import torch

        self.model_name = model_name
        self.device = device or ("cuda" if torch.cuda.is_available() else "cpu")
        self.max_length = max_length
        self.batch_size = batch_size or int(
            os.getenv("LEAN_EXPLORE_EMBEDDING_BATCH_SIZE", DEFAULT_BATCH_SIZE))
        self.query_prompt = query_prompt
        if model is None or tokenizer is None:
            from transformers import AutoModel, AutoTokenizer

            logger.info("Loading embedding model %s on %s", model_name, self.device)
            tokenizer = AutoTokenizer.from_pretrained(model_name, padding_side="left")
            model = AutoModel.from_pretrained(
                model_name, torch_dtype=dtype or (torch.bfloat16 if self.device != "cpu"
                                                  else torch.float32))
        self.tokenizer = tokenizer
        self.model = model.to(self.device).eval()
        if fused_norms:
            from .fused import fuse_rmsnorm

            fuse_rmsnorm(self.model)  # one kernel per RMSNorm instead of six
        from .graphs import GraphRunner

        # hipGraph replay per (batch, padded length) bucket; off by default like the reference
        self._forward = GraphRunner(self._last_token, enabled=use_graphs)

    def _last_token(self, input_ids, attention_mask):
        hidden = self.model(input_ids=input_ids, attention_mask=attention_mask).last_hidden_state
        return hidden[:, -1].float()  # left padding: the last position is the last token

    def encode(self, texts: list[str], is_query: bool = False) -> np.ndarray:
        """float32 [len(texts), d], rows L2-normalised (the pipeline's own Normalize layer)."""
        import torch

        if is_query:  # asymmetric model: only queries carry the prompt (embedding_client.py:84-99)
            texts = [self.query_prompt + t for t in texts]
        out = []
        with torch.no_grad():
            for i in range(0, len(texts), self.batch_size):
                enc = self.tokenizer(texts[i:i + self.batch_size], padding=True, truncation=True,
                                     max_length=self.max_length or 512, return_tensors="pt")
                enc = {k: v.to(self.device) for k, v in enc.items()}
                pooled = self._forward(enc["input_ids"], enc["attention_mask"])
                out.append(torch.nn.functional.normalize(pooled, p=2, dim=1).cpu().numpy())
        return np.concatenate(out, axis=0) if out else np.zeros((0, 0), np.float32)

    async def embed(self, texts: list[str], is_query: bool = False) -> EmbeddingResponse:
        loop = asyncio.get_event_loop()
        embeddings = await loop.run_in_executor(None, self.encode, texts, is_query)
        return EmbeddingResponse(texts=texts, embeddings=[e.tolist() for e in embeddings],
                                 model=self.model_name)
