"""EmbeddingClient / RerankerClient: call surface and invariants of the
reference clients (reference tests/util/embedding_client_test.py, reranker_client_test.py mock the
model and check shapes only). Here a tiny random Qwen3 stands in, on CPU."""

import asyncio

import numpy as np
import pytest

from lean_explore_amd.util import EmbeddingClient, RerankerClient
from lean_explore_amd.util.synthetic import HashTokenizer, random_qwen3

TINY = dict(vocab_size=512, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
            num_attention_heads=4, num_key_value_heads=2, head_dim=16)


def run(coro):
    return asyncio.new_event_loop().run_until_complete(coro)


@pytest.fixture(scope="module")
def embedder():
    return EmbeddingClient("tiny-random-qwen3", device="cpu", max_length=32, batch_size=2,
                           model=random_qwen3(seed=1, **TINY), tokenizer=HashTokenizer(512))


def test_embed_shapes_norms_and_prompt(embedder):
    texts = ["continuous function on a compact set", "prime number", "a b c d e f g"]
    r = run(embedder.embed(texts))
    e = np.array(r.embeddings, dtype=np.float32)
    assert r.texts == texts and r.model == "tiny-random-qwen3" and e.shape == (3, 64)
    assert np.allclose(np.linalg.norm(e, axis=1), 1.0, atol=1e-5)
    q = np.array(run(embedder.embed(texts, is_query=True)).embeddings, dtype=np.float32)
    assert not np.allclose(q, e, atol=1e-3)  # the query prompt changes the encoding


def test_query_prompt_is_what_prompt_name_query_sends():
    """Reference util/embedding_client.py:90-99: `model.encode(texts, prompt_name="query")` only when is_query - sentence-
    transformers then PREPENDS prompts["query"] of the model's config to every text, and nothing to documents
    (Qwen3-Embedding's config_sentence_transformers.json: "Instruct: Given a web search query, retrieve relevant passages
    that answer the query\nQuery:"). Pinned here: the exact strings the tokenizer receives, batch by batch (the
    reference's batch size 8 and LEAN_EXPLORE_EMBEDDING_BATCH_SIZE, :13,49-52), and that embed() passes is_query on."""
    from lean_explore_amd.util.embedding_client import DEFAULT_BATCH_SIZE, QUERY_PROMPT

    assert QUERY_PROMPT == ("Instruct: Given a web search query, retrieve relevant passages that answer the query\nQuery:")
    assert DEFAULT_BATCH_SIZE == 8
    seen = []
    inner = HashTokenizer(512)

    class Recording:
        def __call__(self, texts, **kw):
            seen.append((list(texts), kw.get("padding"), kw.get("truncation"), kw.get("max_length")))
            return inner(texts, **kw)

    emb = EmbeddingClient("tiny-random-qwen3", device="cpu", max_length=32, batch_size=2,
                          model=random_qwen3(seed=1, **TINY), tokenizer=Recording())
    docs = ["continuous function", "prime number", "Nat.add_comm"]
    emb.encode(docs)
    assert [t for t, *_ in seen] == [docs[:2], docs[2:]]  # documents: no prompt, batches of batch_size
    assert all(pad is True and trunc is True and ml == 32 for _, pad, trunc, ml in seen)
    seen.clear()
    run(emb.embed(docs[:1], is_query=True))
    assert [t for t, *_ in seen] == [[QUERY_PROMPT + docs[0]]]  # queries: the config's prompt, prepended verbatim
    seen.clear()
    run(emb.embed(docs[:1]))
    assert [t for t, *_ in seen] == [[docs[0]]]  # is_query defaults to False (documents), as in the reference


def test_embed_is_independent_of_batch_composition(embedder):
    """Left padding + last-token pooling: a text's vector must not depend on its batch mates."""
    texts = ["x y z", "one two three four five six seven eight", "p", "q r"]
    together = embedder.encode(texts)
    alone = np.concatenate([embedder.encode([t]) for t in texts])
    assert np.allclose(together, alone, atol=2e-5)


def test_reranker_scores_and_batching():
    tok = HashTokenizer(512)
    rr = RerankerClient("tiny-random-qwen3", device="cpu", max_length=64, batch_size=2,
                        model=random_qwen3(causal_lm=True, seed=2, **TINY), tokenizer=tok)
    docs = ["theorem about groups", "lemma on compactness", "definition of a ring", "x"]
    r = run(rr.rerank("compact", docs))  # 4 > batch_size: executor batches
    assert r.query == "compact" and len(r.scores) == 4 and all(0.0 <= s <= 1.0 for s in r.scores)
    s_sync = rr.rerank_sync("compact", docs).scores  # one batch of 4
    assert np.allclose(r.scores, s_sync, atol=2e-5)
    assert run(rr.rerank("compact", [])).scores == []
    assert rr._format_pair("q", "d") == (
        "<Instruct>: Find relevant Lean 4 math declarations\n<Query>: q\n<Document>: d")


@pytest.mark.gpu
def test_graph_replay_matches_eager_on_gpu():
    """hipGraph replay (bucketed left padding, filler rows) against the eager forward."""
    import torch

    tok = HashTokenizer(512)
    texts = ["a b c", "one two three four five six seven eight nine ten eleven", "p q"]
    outs = []
    for graphs in (False, True):
        emb = EmbeddingClient("tiny", device="cuda", max_length=64, batch_size=4, use_graphs=graphs,
                              model=random_qwen3(seed=1, dtype=torch.float32, **TINY), tokenizer=tok)
        outs.append((emb.encode(texts), emb.encode(texts[:1], is_query=True), emb._forward.replays))
    assert outs[0][2] == 0 and outs[1][2] >= 2  # the graphed client really replayed
    assert np.allclose(outs[0][0], outs[1][0], atol=1e-4)
    assert np.allclose(outs[0][1], outs[1][1], atol=1e-4)
    scores = []
    for graphs in (False, True):
        rr = RerankerClient("tiny", device="cuda", max_length=64, batch_size=4, use_graphs=graphs,
                            model=random_qwen3(causal_lm=True, seed=2, dtype=torch.float32, **TINY),
                            tokenizer=tok, token_true_id=5, token_false_id=9)
        scores.append(run(rr.rerank("compact", ["d one", "d two two", "d3", "d 4 4 4", "d5", "d6"])).scores)
    assert np.allclose(scores[0], scores[1], atol=1e-4)


def test_fused_rmsnorm_matches_the_eager_modules():
    """`fuse_rmsnorm` re-points HF's six-kernel RMSNorm at torch.nn.functional.rms_norm: the
    embeddings / rerank scores must agree with the untouched modules (fp32: to rounding; bf16: to
    the dtype's resolution), and every RMSNorm of the model must have been switched."""
    import torch

    from lean_explore_amd.util.fused import fuse_rmsnorm

    texts = ["continuous function on a compact set", "prime number", "a b c d e f g"]
    for dtype, tol in ((torch.float32, 1e-5), (torch.bfloat16, 3e-2)):
        plain = EmbeddingClient("t", device="cpu", max_length=32, tokenizer=HashTokenizer(512),
                                model=random_qwen3(seed=3, dtype=dtype, **TINY), fused_norms=False)
        fused = EmbeddingClient("t", device="cpu", max_length=32, tokenizer=HashTokenizer(512),
                                model=random_qwen3(seed=3, dtype=dtype, **TINY), fused_norms=True)
        a, b = plain.encode(texts, is_query=True), fused.encode(texts, is_query=True)
        assert np.abs(a - b).max() <= tol, (dtype, np.abs(a - b).max())
    m = random_qwen3(seed=4, **TINY)
    n_norms = sum(1 for x in m.modules() if type(x).__name__.endswith("RMSNorm"))
    assert fuse_rmsnorm(m) == n_norms == 2 * 4 + 1   # 4 per layer (input, post-attn, q, k) + final
    rr_a = RerankerClient("t", device="cpu", max_length=64, tokenizer=HashTokenizer(512),
                          token_true_id=1, token_false_id=2, fused_norms=False,
                          model=random_qwen3(causal_lm=True, seed=5, **TINY))
    rr_b = RerankerClient("t", device="cpu", max_length=64, tokenizer=HashTokenizer(512),
                          token_true_id=1, token_false_id=2, fused_norms=True,
                          model=random_qwen3(causal_lm=True, seed=5, **TINY))
    docs = ["sum of zero", "unrelated words here", "zero zero sum and more"]
    sa = rr_a.rerank_sync("sum and zero", docs).scores
    sb = rr_b.rerank_sync("sum and zero", docs).scores
    assert np.allclose(sa, sb, atol=1e-5)
