"""Service.search() end to end on the MI355X: index file on disk -> read_index -> HIP search ->
row/id mapping -> fusion, against the same engine driven by the CPU oracle."""

import numpy as np
import pytest

from lean_explore_amd import faiss_compat, loader
from lean_explore_amd import search as S
from tests import helpers as H
from tests.test_glue_cpu import FakeEmbed, OracleIndex, _make_db, run

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("container", ["flat", "ivf"])
def test_service_search_drop_in(tmp_path, container):
    import json

    n, d = 3000, 1024
    corpus = H.gauss(17, n, d)
    rows = []
    for i in range(n):
        deps = json.dumps([f"Mathlib.decl{(i * 7 + 1) % n}"]) if i % 4 == 0 else None
        rows.append((5000 + i, f"Mathlib.decl{i}" if i % 50 else f"Mathlib.S{i}.mk", "Mathlib.Mod",
                     "doc", f"theorem t{i}", f"http://x/{i}", deps, f"statement {i}",
                     loader.embedding_to_blob(corpus[i].tolist())))
    db = tmp_path / "lean_explore.db"
    _make_db(db, rows)
    ids, loaded = loader.load_corpus_from_sqlite(db)
    assert np.array_equal(loaded, corpus)
    loader.save_ids_map(tmp_path / "informalization_faiss_ids_map.json", ids)
    index_path = tmp_path / "informalization_faiss.index"
    if container == "flat":
        ix = faiss_compat.IndexFlatIP(d)
        ix.add(loaded)
        faiss_compat.write_index(ix, index_path)
    else:  # the container the reference ships (IVF-flat), searched exactly
        assign = np.random.default_rng(3).integers(0, 256, size=n)
        H.write_ivf_flat(index_path, loaded, assign, 256)

    qvec = corpus[123] * 3.0 + 0.01 * H.gauss(5, 1, d)[0]
    hip = S.SearchEngine(base_path=tmp_path, embedding_client=FakeEmbed(qvec), lexical_retriever=False)
    ref = S.SearchEngine(db_path=db, embedding_client=FakeEmbed(qvec), index=OracleIndex(loaded),
                         ids_map=ids, lexical_retriever=False)
    sem_hip = run(hip._retrieve_semantic_candidates("q", 1000))
    sem_ref = run(ref._retrieve_semantic_candidates("q", 1000))
    assert list(sem_hip) == list(sem_ref)                       # same ids in the same rank order
    assert list(sem_hip)[0] == 5123
    assert np.allclose(list(sem_hip.values()), list(sem_ref.values()), atol=1e-5)

    r_hip = run(S.Service(engine=hip).search("q", limit=20, rerank_top=0))
    r_ref = run(S.Service(engine=ref).search("q", limit=20, rerank_top=0))
    assert [r.id for r in r_hip.results] == [r.id for r in r_ref.results]
    assert r_hip.count == len(r_hip.results) > 0
    assert all(not r.name.endswith(".mk") for r in r_hip.results)


def test_hybrid_search_with_lexical_retriever(tmp_path):
    """Both retrievers on the GPU, fused by reciprocal rank as in engine.py:263-300."""
    import json

    from lean_explore_amd.bm25 import NameRetriever

    n, d = 2000, 384
    corpus = H.gauss(23, n, d)
    names = [f"Mathlib.lemma_{i}_add_comm" if i % 2 else f"Mathlib.thm{i}.mul_zero" for i in range(n)]
    rows = [(7000 + i, names[i], "Mathlib.Mod", None, "src", "link", None, f"text {i}",
             loader.embedding_to_blob(corpus[i].tolist())) for i in range(n)]
    db = tmp_path / "lean_explore.db"
    _make_db(db, rows)
    ids, loaded = loader.load_corpus_from_sqlite(db)
    ix = faiss_compat.IndexFlatIP(d)
    ix.add(loaded)
    lexical = NameRetriever.from_names(ids, names)
    eng = S.SearchEngine(db_path=db, embedding_client=FakeEmbed(corpus[400]), index=ix, ids_map=ids,
                         lexical_retriever=lexical)
    res = run(eng.search("thm401 mul_zero", limit=10, rerank_top=0))
    assert res and all(isinstance(r, S.SearchResult) for r in res)
    sem = run(eng._retrieve_semantic_candidates("q", 1000))
    lex = eng._retrieve_bm25_candidates("thm401 mul_zero", 1000)
    fused = dict(S.SearchEngine._compute_rrf_scores(lex, sem))
    best = max(fused, key=fused.get)
    assert res[0].id == best or fused[res[0].id] == pytest.approx(fused[best])
    assert 7400 in sem and len(lex) > 0


def test_rerank_mixing_weights(tmp_path):
    """engine.py:360-416: 1.0*reranker + 0.4*bm25(informalization) + 0.2*deps (+1.0*fuzzy>=0.7)."""
    from types import SimpleNamespace

    class FakeReranker:
        async def rerank(self, query, documents):
            self.documents = documents
            return SimpleNamespace(scores=[float(len(d) % 7) for d in documents])

    n, d = 300, 64
    corpus = H.gauss(29, n, d)
    rows = [(100 + i, f"Pkg.decl{i}", "Pkg.Mod", None, "src", "link", None,
             f"the sum of {i} and zero equals {i}" if i % 2 else f"a list of length {i}",
             loader.embedding_to_blob(corpus[i].tolist())) for i in range(n)]
    db = tmp_path / "lean_explore.db"
    _make_db(db, rows)
    ids, loaded = loader.load_corpus_from_sqlite(db)
    ix = faiss_compat.IndexFlatIP(d)
    ix.add(loaded)
    rr = FakeReranker()
    eng = S.SearchEngine(db_path=db, embedding_client=FakeEmbed(corpus[10]), index=ix, ids_map=ids,
                         reranker_client=rr, lexical_retriever=False)
    res = run(eng.search("sum and zero", limit=5, rerank_top=20))
    assert len(res) == 5 and len(rr.documents) == 20
    assert rr.documents[0].startswith("Pkg.decl")            # "name: informalization"
    decls = [SimpleNamespace(name=f"n{i}", informalization=t, dependencies=None)
             for i, t in enumerate(["sum of zero", "unrelated words", "zero zero sum and"])]
    b = eng._compute_bm25_on_informalizations("sum and zero", [(d_, 0.0) for d_ in decls])
    assert b[2] > b[0] > b[1]


def test_engine_loads_bm25_indices_from_base_path(tmp_path):
    """Like the reference (engine.py:175-190), SearchEngine(base_path=...) picks up the BM25 name
    indices saved under the reference's file names without being handed a retriever."""
    from lean_explore_amd.bm25 import NameRetriever

    n, d = 1500, 128
    corpus = H.gauss(31, n, d)
    names = [f"Mathlib.lemma_{i}_add_comm" if i % 2 else f"Mathlib.thm{i}.mul_zero" for i in range(n)]
    rows = [(7000 + i, names[i], "Mathlib.Mod", None, "src", "link", None, f"text {i}",
             loader.embedding_to_blob(corpus[i].tolist())) for i in range(n)]
    _make_db(tmp_path / "lean_explore.db", rows)
    ids, loaded = loader.load_corpus_from_sqlite(tmp_path / "lean_explore.db")
    loader.save_ids_map(tmp_path / "informalization_faiss_ids_map.json", ids)
    ix = faiss_compat.IndexFlatIP(d)
    ix.add(loaded)
    faiss_compat.write_index(ix, tmp_path / "informalization_faiss.index")
    built = NameRetriever.from_names(ids, names)
    built.save(tmp_path)
    eng = S.SearchEngine(base_path=tmp_path, embedding_client=FakeEmbed(corpus[10]))
    lex = eng._retrieve_bm25_candidates("thm400 mul_zero", 1000)
    assert lex == built("thm400 mul_zero", 1000) and 7400 in lex
    res = run(eng.search("thm400 mul_zero", limit=5, rerank_top=0))
    assert len(res) == 5


def test_concurrent_service_searches_share_corpus_passes(tmp_path):
    """Round 6 (verdict item 4): with ``concurrent_dense=True`` the dense call leaves the event-loop thread, so
    concurrent Service.search() coroutines reach ls_search together and are served as combined batches (debug
    counter 16) - and every coroutine gets exactly the result the serial, reference-shaped engine gives."""
    import asyncio

    n, d = 20_000, 384
    corpus = H.gauss(37, n, d)
    rows = [(9000 + i, f"Pkg.decl{i}", "Pkg.Mod", None, "src", "link", None, f"text {i}",
             loader.embedding_to_blob(corpus[i].tolist())) for i in range(n)]
    db = tmp_path / "lean_explore.db"
    _make_db(db, rows)
    ids, loaded = loader.load_corpus_from_sqlite(db)
    ix = faiss_compat.IndexFlatIP(d)
    ix.add(loaded)

    class PerQueryEmbed:  # the query text names the corpus row its vector is near
        async def embed(self, texts, is_query=False):
            from types import SimpleNamespace

            await asyncio.sleep(0)  # (a real client awaits its executor here: the coroutines interleave)
            v = corpus[int(texts[0])] * 2.0 + 0.01 * corpus[(int(texts[0]) * 7 + 1) % n]
            return SimpleNamespace(embeddings=[v.tolist()])

    serial = S.Service(engine=S.SearchEngine(db_path=db, embedding_client=PerQueryEmbed(), index=ix, ids_map=ids,
                                             lexical_retriever=False))
    conc_engine = S.SearchEngine(db_path=db, embedding_client=PerQueryEmbed(), index=ix, ids_map=ids,
                                 lexical_retriever=False, concurrent_dense=True)
    conc = S.Service(engine=conc_engine)
    queries = [str(100 * j + 3) for j in range(8)]
    want = [[r.id for r in run(serial.search(q, limit=20, rerank_top=0)).results] for q in queries]
    hip = conc_engine.faiss_informal_index  # (faiss_compat.IndexFlatIP is a FlatIPIndex)
    before = hip.debug_counter(16)

    async def many():
        outs = []
        for _ in range(30):  # (30 rounds of 8 concurrent searches: some of them must meet in the library)
            outs = await asyncio.gather(*[conc.search(q, limit=20, rerank_top=0) for q in queries])
        return outs

    got = run(many())
    assert [[r.id for r in resp.results] for resp in got] == want
    assert want[0][0] == 9003 and hip.debug_counter(16) > before, "no two searches were combined"
