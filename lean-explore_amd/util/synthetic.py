"""Stand-ins for what cannot be downloaded here: a random-initialised Qwen3 of a given shape
(the production checkpoints Qwen3-Embedding-0.6B / Qwen3-Reranker-0.6B share one architecture:
28 layers, hidden 1024, 16 heads / 8 KV heads of 128, MLP 3072, vocab 151 669) and a
deterministic hashing tokenizer with the HF call signature. Used by tests and by
tools/hybrid_bench.py; results are meaningless as retrieval quality, the compute is the real
shape."""

from __future__ import annotations

import zlib

QWEN3_06B = dict(vocab_size=151669, hidden_size=1024, intermediate_size=3072,
                 num_hidden_layers=28, num_attention_heads=16, num_key_value_heads=8,
                 head_dim=128, max_position_embeddings=32768)


def random_qwen3(causal_lm: bool = False, seed: int = 0, dtype=None, **overrides):
    import torch
    from transformers import Qwen3Config, Qwen3ForCausalLM, Qwen3Model

    cfg = Qwen3Config(**{**QWEN3_06B, **overrides})
    torch.manual_seed(seed)
    model = (Qwen3ForCausalLM if causal_lm else Qwen3Model)(cfg)
    return model.to(dtype) if dtype is not None else model


class HashTokenizer:
    """Whitespace pieces -> crc32 % vocab; left padding; HF-style call."""

    def __init__(self, vocab_size: int, pad_id: int = 0):
        self.vocab_size, self.pad_id = vocab_size, pad_id

    def convert_tokens_to_ids(self, token: str) -> int:
        return 1 + zlib.crc32(token.encode()) % (self.vocab_size - 1)

    def __call__(self, texts, padding=True, truncation=True, max_length=512, return_tensors="pt"):
        import torch

        ids = [[self.convert_tokens_to_ids(t) for t in s.split()][:max_length] or [1]
               for s in texts]
        width = max(len(x) for x in ids)
        input_ids = torch.full((len(ids), width), self.pad_id, dtype=torch.long)
        mask = torch.zeros((len(ids), width), dtype=torch.long)
        for r, x in enumerate(ids):
            input_ids[r, width - len(x):] = torch.tensor(x)
            mask[r, width - len(x):] = 1
        return {"input_ids": input_ids, "attention_mask": mask}
