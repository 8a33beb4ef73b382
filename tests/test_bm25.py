"""Lexical (BM25+) retrieval, SURVEY §8(f) row 3: numpy oracle vs the textbook formula and the
product's host builder on CPU; HIP retrieval vs the oracle bit for bit on the GPU."""

import json

import numpy as np
import pytest

from lean_explore_amd.bm25 import BM25Index, NameRetriever
from lean_explore_amd.search.tokenization import tokenize_raw, tokenize_spaced
from oracle import bm25_ref as R

WORDS = ["nat", "add", "comm", "mul", "zero", "succ", "list", "map", "append", "real", "sqrt", "le",
         "lt", "of", "iff", "continuous", "on", "measure", "theory", "group", "ring", "hom", "ker"]


def synth_names(n, seed=0):
    rng = np.random.default_rng(seed)
    names = []
    for i in range(n):
        parts = [WORDS[j] for j in rng.integers(0, len(WORDS), size=rng.integers(1, 5))]
        ns = ["Nat", "List", "Real", "MeasureTheory", "Mathlib"][rng.integers(0, 5)]
        names.append(f"{ns}.{'_'.join(parts)}{i % 97 if i % 3 == 0 else ''}")
    return names


def test_oracle_matches_textbook_bm25_plus():
    corpus = [tokenize_spaced(n) for n in synth_names(300, 1)]  # repeated tokens: tf > 1 occurs
    ix = R.build(corpus)
    for q in (["nat", "add"], ["list", "map", "append", "list"], ["nope"], []):
        s = R.scores(ix, R.token_ids(ix, q))
        b = R.brute_force_scores(corpus, q)
        assert np.allclose(s, b, atol=2e-5), q
    docs, sc = R.retrieve(ix, ["nat", "add"], 400)
    assert (docs[300:] == -1).all() and (np.diff(sc[:300]) <= 0).all()


def test_host_builder_equals_oracle_and_roundtrips(tmp_path):
    corpus = [list(dict.fromkeys(tokenize_spaced(n))) for n in synth_names(500, 2)]
    ref = R.build(corpus)
    ix = BM25Index().index(corpus)
    assert ix.vocab == ref["vocab"] and ix.num_docs == 500
    assert np.array_equal(ix.indptr, ref["indptr"]) and np.array_equal(ix.indices, ref["indices"])
    assert np.array_equal(ix.data, ref["data"]) and np.array_equal(ix.nonoccurrence, ref["nonocc"])
    ix.save(tmp_path / "bm25_name_spaced")
    # the reference's file set (cli/data_commands.py:42-59)
    assert sorted(p.name for p in (tmp_path / "bm25_name_spaced").iterdir()) == sorted([
        "data.csc.index.npy", "indices.csc.index.npy", "indptr.csc.index.npy",
        "nonoccurrence_array.index.npy", "params.index.json", "vocab.index.json"])
    back = BM25Index.load(tmp_path / "bm25_name_spaced")
    assert back.vocab == ix.vocab and np.array_equal(back.data, ix.data)
    assert np.array_equal(back.indptr, ix.indptr) and back.num_docs == 500
    assert json.loads((tmp_path / "bm25_name_spaced" / "params.index.json").read_text())["method"] == "bm25+"


def test_host_scores_of_a_small_index_equal_the_oracle():
    """`get_scores_host` (the throw-away index over the rerank candidates, reference
    search/engine.py:418-448) = bm25s's get_scores: bit-identical to the oracle's sequential sums."""
    corpus = [tokenize_spaced(n) for n in synth_names(60, 7)]
    ref, ix = R.build(corpus), BM25Index().index(corpus)
    for q in (["nat", "add"], ["list", "map", "list"], ["nope"], []):
        assert np.array_equal(ix.get_scores_host(q), R.scores(ref, R.token_ids(ref, q))), q


@pytest.mark.gpu
def test_hip_retrieve_bit_exact_vs_oracle():
    names = synth_names(200_000, 3)
    corpus = [list(dict.fromkeys(tokenize_spaced(n))) for n in names]
    ref = R.build(corpus)
    ix = BM25Index().index(corpus)
    for q, k in ((["nat", "add", "comm"], 1000), (["measure", "theory"], 50), (["nope"], 10),
                 (["list", "list", "map"], 1000), ([], 5), (["ker"], 2048)):
        docs, sc = ix.retrieve(q, k)
        dref, sref = R.retrieve(ref, q, k)
        assert np.array_equal(sc, sref), q      # same float32 accumulation order: bit-identical
        assert np.array_equal(docs, dref), q    # ties (thousands) broken by ascending document
    # tie-heavy queries (discrete BM25 scores: the k-th score is shared by thousands of names) stay
    # on the selection's fast path: rows are dealt to workgroups in granules of 4
    assert ix.debug_counter(0) <= 1 and ix.debug_counter(1) == 0
    # a query of more than 32 tokens is scored by chained launches (partial sums in F), in order
    long_q = (["nat", "add", "comm", "list", "map", "measure", "theory", "ker"] * 9)[:70]
    docs, sc = ix.retrieve(long_q, 300)
    dref, sref = R.retrieve(ref, long_q, 300)
    assert np.array_equal(sc, sref) and np.array_equal(docs, dref)
    # documents with more tokens than the 8 entries a lane keeps in registers
    rng = np.random.default_rng(5)
    fat = [list(dict.fromkeys(WORDS[j] for j in rng.integers(0, len(WORDS), size=rng.integers(1, 40))))
           for _ in range(5000)]
    fx, fref = BM25Index().index(fat), R.build(fat)
    for q in (WORDS[:5], [WORDS[3]] * 3 + WORDS[10:14]):
        docs, sc = fx.retrieve(q, 500)
        dref, sref = R.retrieve(fref, q, 500)
        assert np.array_equal(sc, sref) and np.array_equal(docs, dref), q
    # the host scorer of small throw-away indices returns the very bits the GPU path returns
    tiny = BM25Index().index(corpus[:50])
    docs, sc = tiny.retrieve(["nat", "add", "comm"], 50)
    host = tiny.get_scores_host(["nat", "add", "comm"])
    assert np.array_equal(host[docs], sc)
    small = BM25Index().index(corpus[:30])
    docs, sc = small.retrieve(["nat"], 100)     # k > n_docs -> padding
    dref, sref = R.retrieve(R.build(corpus[:30]), ["nat"], 100)
    assert np.array_equal(docs, dref) and np.array_equal(sc, sref)


@pytest.mark.gpu
def test_name_retriever_max_merge_and_engine_plug(tmp_path):
    names = synth_names(5000, 4)
    ids = [10_000 + i for i in range(len(names))]
    nr = NameRetriever.from_names(ids, names)
    nr.save(tmp_path)
    nr2 = NameRetriever.load(tmp_path)
    q = names[1234]
    got = nr(q, 1000)
    assert got == nr2(q, 1000)
    # restate engine.py:192-223 with the oracle
    want: dict[int, float] = {}
    for toks, corp in ((tokenize_spaced(q), [list(dict.fromkeys(tokenize_spaced(n))) for n in names]),
                       (tokenize_raw(q), [list(dict.fromkeys(tokenize_raw(n))) for n in names])):
        docs, sc = R.retrieve(R.build(corp), toks, 1000)
        for d, s in zip(docs, sc):
            if d >= 0:
                want[ids[d]] = max(want.get(ids[d], 0.0), float(s))
    assert got == want
    assert max(got, key=got.get) == ids[1234] or got[ids[1234]] == max(got.values())
