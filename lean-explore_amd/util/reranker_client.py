"""Cross-encoder reranking on PyTorch-ROCm.

Call surface and scoring rule of the reference's client (reference
src/lean_explore/util/reranker_client.py): pairs are formatted
"<Instruct>: ...\\n<Query>: ...\\n<Document>: ..." (:99-108); the score of a pair is
softmax([logit_false, logit_true])[1] at the last position of a causal LM (:110-141); small
inputs run synchronously, larger ones in executor batches (:164-206). Only the last position's
logits are materialised (`logits_to_keep=1`): the reference computes the full [batch, seq, vocab]
tensor and discards all but one row.

`model` / `tokenizer` / token ids can be injected (random-initialised Qwen3 + synthetic tokenizer
in tests and tools/hybrid_bench.py); otherwise the named checkpoint is loaded.
"""

from __future__ import annotations

import asyncio
import logging
import os
from typing import Any

from pydantic import BaseModel

logger = logging.getLogger(__name__)

DEFAULT_INSTRUCTION = "Find relevant Lean 4 math declarations"  # reranker_client.py:13
DEFAULT_CUDA_BATCH_SIZE = 16
DEFAULT_CPU_BATCH_SIZE = 32


class RerankerResponse(BaseModel):
    """reference reranker_client.py:18-28"""

    query: str
    scores: list[float]
    model: str


class RerankerClient:
    def __init__(self, model_name: str = "Qwen/Qwen3-Reranker-0.6B", device: str | None = None,
                 max_length: int = 512, instruction: str = DEFAULT_INSTRUCTION,
                 batch_size: int | None = None, *, model: Any = None, tokenizer: Any = None,
                 token_true_id: int | None = None, token_false_id: int | None = None,
                 dtype: Any = None, use_graphs: bool = False, fused_norms: bool = False):
        import torch

        self.model_name = model_name
        self.device = device or ("cuda" if torch.cuda.is_available() else "cpu")
        self.max_length = max_length
        self.instruction = instruction
        env = os.getenv("LEAN_EXPLORE_RERANKER_BATCH_SIZE")
        self.batch_size = batch_size or (int(env) if env else (
            DEFAULT_CUDA_BATCH_SIZE if self.device == "cuda" else DEFAULT_CPU_BATCH_SIZE))
        if model is None or tokenizer is None:
            from transformers import AutoModelForCausalLM, AutoTokenizer

            logger.info("Loading reranker model %s on %s", model_name, self.device)
            tokenizer = AutoTokenizer.from_pretrained(model_name, padding_side="left")
            model = AutoModelForCausalLM.from_pretrained(
                model_name, torch_dtype=dtype or (torch.float16 if self.device == "cuda"
                                                  else torch.float32))
        self.tokenizer = tokenizer
        self.model = model.to(self.device).eval()
        if fused_norms:
            from .fused import fuse_rmsnorm

            fuse_rmsnorm(self.model)  # one kernel per RMSNorm instead of six
        from .graphs import GraphRunner

        self._forward = GraphRunner(self._last_logits, batch_step=self.batch_size,
                                    enabled=use_graphs)
        self._token_true_id = (token_true_id if token_true_id is not None
                               else tokenizer.convert_tokens_to_ids("true"))
        self._token_false_id = (token_false_id if token_false_id is not None
                                else tokenizer.convert_tokens_to_ids("false"))

    def _last_logits(self, input_ids, attention_mask):
        return self.model(input_ids=input_ids, attention_mask=attention_mask,
                          logits_to_keep=1).logits[:, -1, :].float()

    def _format_pair(self, query: str, document: str) -> str:
        return f"<Instruct>: {self.instruction}\n<Query>: {query}\n<Document>: {document}"

    def _compute_scores_sync(self, pairs: list[str]) -> list[float]:
        import torch

        with torch.no_grad():
            enc = self.tokenizer(pairs, padding=True, truncation=True, max_length=self.max_length,
                                 return_tensors="pt")
            enc = {k: v.to(self.device) for k, v in enc.items()}
            logits = self._forward(enc["input_ids"], enc["attention_mask"])
            stacked = torch.stack([logits[:, self._token_false_id],
                                   logits[:, self._token_true_id]], dim=1)
            return torch.nn.functional.log_softmax(stacked, dim=1)[:, 1].exp().cpu().tolist()

    def rerank_sync(self, query: str, documents: list[str]) -> RerankerResponse:
        if not documents:
            return RerankerResponse(query=query, scores=[], model=self.model_name)
        pairs = [self._format_pair(query, d) for d in documents]
        return RerankerResponse(query=query, scores=self._compute_scores_sync(pairs),
                                model=self.model_name)

    async def rerank(self, query: str, documents: list[str], batch_size: int | None = None
                     ) -> RerankerResponse:
        if not documents:
            return RerankerResponse(query=query, scores=[], model=self.model_name)
        batch_size = batch_size or self.batch_size
        if len(documents) <= batch_size:  # small inputs skip the executor hop
            return self.rerank_sync(query, documents)
        pairs = [self._format_pair(query, d) for d in documents]
        loop = asyncio.get_event_loop()
        scores: list[float] = []
        for i in range(0, len(pairs), batch_size):
            scores.extend(await loop.run_in_executor(None, self._compute_scores_sync,
                                                     pairs[i:i + batch_size]))
        return RerankerResponse(query=query, scores=scores, model=self.model_name)
