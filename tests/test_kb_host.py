"""CPU: host logic of the retrieve() path (KB / AsyncKB mirror, matrix cache and
its invalidation, attach()) against the transcript captured from the real
reference (tests/golden/kb_cases.json).  The arithmetic is a test double
(tests/fake_backend.py); the same transcript runs on the HIP path in
tests/test_kb_gpu.py."""
import asyncio
import json
import os
import sys

import numpy as np
import pytest

from fake_backend import OracleIndex
from kb_transcript import embedding_func_from, load_cases, replay_async, replay_sync

import svs_amd
from svs_amd import matrix as mx


def test_sync_kb_transcript(tmp_path):
    cases = load_cases()
    kb = svs_amd.KB(str(tmp_path / "a.sqlite"), embedding_func_from(cases), index_factory=OracleIndex)
    replay_sync(kb, cases)
    kb.close()
    assert OracleIndex.live == 0


def test_async_kb_transcript(tmp_path):
    cases = load_cases()

    async def run():
        kb = svs_amd.AsyncKB(str(tmp_path / "b.sqlite"), embedding_func_from(cases), index_factory=OracleIndex)
        await replay_async(kb, cases)
        await kb.close()

    asyncio.run(run())
    assert OracleIndex.live == 0


def test_retrieve_many_equals_loop_of_retrieve(tmp_path):
    rng = np.random.default_rng(3)
    vecs = rng.standard_normal((120, 16)); vecs /= np.linalg.norm(vecs, axis=1, keepdims=True)
    table = {f"doc {i}": [float(x) for x in vecs[i]] for i in range(120)}

    async def ef(texts):
        return [table[t] for t in texts]

    kb = svs_amd.KB(str(tmp_path / "many.sqlite"), ef, index_factory=OracleIndex)
    with kb.bulk_add_docs() as add_doc:
        for i in range(100):
            add_doc(f"doc {i}", meta={"i": i})
    qs = [f"doc {i}" for i in (5, 17, 99, 101, 119)]
    many = kb.retrieve_many(qs, 7)
    assert many == [kb.retrieve(q, 7) for q in qs]
    assert kb.retrieve_many([], 3) == []
    kb.close()


def test_document_top_pairwise_scores(tmp_path):
    """Reference tests/test_kb.py:1788-1797: after adding third/first/second doc the two
    best pairs are (1,2) then (2,3)."""
    cases = load_cases()
    kb = svs_amd.KB(str(tmp_path / "p.sqlite"), embedding_func_from(cases), index_factory=OracleIndex)
    with kb.bulk_add_docs() as add_doc:
        for t in ("third doc", "first doc", "second doc"):
            add_doc(t)
    recs = kb.document_top_pairwise_scores(n=2)
    assert [(a["id"], b["id"]) for _, a, b in recs] == [(1, 2), (2, 3)]
    assert len(kb.document_top_pairwise_scores(n=100)) == 3
    kb.close()


def test_async_retrieve_many_and_top_pairs(tmp_path):
    """AsyncKB twins (reference src/svs/kb.py:1208-1243 for the pairs; retrieve_many is SURVEY 8(f)
    rank 3): equal to the sync KB on the same file, element by element, and safe to run concurrently
    with each other and with an add (the searches hold their own reference to the matrix)."""
    rng = np.random.default_rng(7)
    vecs = rng.standard_normal((150, 16)); vecs /= np.linalg.norm(vecs, axis=1, keepdims=True)
    table = {f"doc {i}": [float(x) for x in vecs[i]] for i in range(150)}

    async def ef(texts):
        return [table[t] for t in texts]

    path = str(tmp_path / "amany.sqlite")
    kb = svs_amd.KB(path, ef, index_factory=OracleIndex)
    with kb.bulk_add_docs() as add_doc:
        for i in range(100):
            add_doc(f"doc {i}", meta={"i": i})
    qs = [f"doc {i}" for i in (5, 17, 99, 101, 149)]
    want_many = kb.retrieve_many(qs, 7)
    want_pairs = kb.document_top_pairwise_scores(12)
    kb.close()

    async def run():
        akb = svs_amd.AsyncKB(path, ef, index_factory=OracleIndex)
        got_many, got_pairs, one = await asyncio.gather(akb.retrieve_many(qs, 7), akb.document_top_pairwise_scores(12),
                                                        akb.retrieve(qs[0], 7))
        assert got_many == want_many and one == want_many[0]
        assert [(s, a["id"], b["id"]) for s, a, b in got_pairs] == [(s, a["id"], b["id"]) for s, a, b in want_pairs]
        assert await akb.retrieve_many([], 3) == []
        # the reference's own pair test (tests/test_kb.py:1254-1263 async twin): counts clamp
        assert len(await akb.document_top_pairwise_scores(10 ** 6)) == 100 * 99 // 2
        async with akb.bulk_add_docs() as add_doc:          # an add between two batched searches
            await add_doc("doc 120")
        after = await akb.retrieve_many(["doc 120", "doc 5"], 3)
        assert after[0][0]["doc"]["text"] == "doc 120" and abs(after[0][0]["score"] - 1.0) < 1e-5
        await akb.close()

    asyncio.run(run())
    assert OracleIndex.live == 0


def test_incremental_add_and_delete_edit_the_loaded_matrix(tmp_path):
    """bulk_add_docs / bulk_del_docs on a LOADED KB append / tombstone rows of the
    HBM copy instead of dropping it; results must equal a KB
    rebuilt from storage."""
    rng = np.random.default_rng(12)
    vecs = rng.standard_normal((300, 24)); vecs /= np.linalg.norm(vecs, axis=1, keepdims=True)
    table = {f"doc {i}": [float(x) for x in vecs[i]] for i in range(300)}

    async def ef(texts):
        return [table[t] for t in texts]

    path = str(tmp_path / "inc.sqlite")
    kb = svs_amd.KB(path, ef, index_factory=OracleIndex)
    with kb.bulk_add_docs() as add_doc:
        for i in range(100):
            add_doc(f"doc {i}")
    kb.load()
    first_index = kb.embeddings_matrix.index
    with kb.bulk_add_docs() as add_doc:                 # append
        for i in range(100, 180):
            add_doc(f"doc {i}")
    assert kb.embeddings_matrix.index is first_index and first_index.shape[0] == 180
    with kb.bulk_del_docs() as del_doc:                 # tombstone (doc id == i + 1)
        for i in (3, 50, 120, 179):
            del_doc(i + 1)
    assert kb.embeddings_matrix.index is first_index     # still the same HBM copy
    with kb.bulk_add_docs() as add_doc:
        add_doc("doc 200")
    fresh = svs_amd.KB(path, ef, index_factory=OracleIndex)   # rebuilt from storage
    for q in ("doc 3", "doc 120", "doc 200", "doc 17", "doc 250"):
        a, b = kb.retrieve(q, 12), fresh.retrieve(q, 12)
        assert [(d["doc"]["id"], d["score"]) for d in a] == [(d["doc"]["id"], d["score"]) for d in b], q
    assert len(kb.retrieve("doc 1", 1000)) == 177
    assert [(s, a["id"], b["id"]) for s, a, b in kb.document_top_pairwise_scores(20)] == \
           [(s, a["id"], b["id"]) for s, a, b in fresh.document_top_pairwise_scores(20)]
    # a failed transaction leaves storage untouched and simply drops the cache
    with pytest.raises(KeyError):
        with kb.bulk_del_docs() as del_doc:
            del_doc(5); del_doc(99999)
    assert [d["doc"]["id"] for d in kb.retrieve("doc 4", 3)] == [d["doc"]["id"] for d in fresh.retrieve("doc 4", 3)]
    # many deletes trigger compaction (rebuild) instead of ever more tombstones
    with kb.bulk_del_docs() as del_doc:
        for i in range(60, 110):
            del_doc(i + 1)
    fresh.close(); fresh = svs_amd.KB(path, ef, index_factory=OracleIndex)
    assert [d["doc"]["id"] for d in kb.retrieve("doc 70", 5)] == [d["doc"]["id"] for d in fresh.retrieve("doc 70", 5)]
    kb.close(); fresh.close()
    assert OracleIndex.live == 0


def test_matrix_build_kat(tmp_path):
    """A7: BLOB rows -> (matrix, lookup), non-contiguous ids after a delete
    (reference tests/test_kb.py:753-806)."""
    from svs_amd.kb import _Store
    g = load_cases()["matrix_build"]
    st = _Store(str(tmp_path / "m.sqlite"))
    with st.transaction():
        for i, b in enumerate(g[0]["blobs_hex"]):
            st.set_doc_embedding(st.add_doc(f"doc {i}", None, None), bytes.fromhex(b))
        m, lk = st.build_embeddings_matrix()
        assert m.dtype == np.float32 and m.flags["C_CONTIGUOUS"] and lk.dtype == np.int64
        assert m.tolist() == g[0]["matrix"] and lk.tolist() == g[0]["lookup"]
        st.del_doc(g[1]["deleted_doc"])
        m, lk = st.build_embeddings_matrix()
        assert m.tolist() == g[1]["matrix"] and lk.tolist() == g[1]["lookup"]
    st.close()


def test_codec_kat():
    from svs_amd.kb import embedding_from_bytes, embedding_to_bytes
    for c in load_cases()["codec"]:
        assert embedding_to_bytes(c["values"]).hex() == c["hex"]
        assert np.allclose(embedding_from_bytes(bytes.fromhex(c["hex"])), np.array(c["values"], dtype=np.float32))


def test_magnitude_guard_and_rollback(tmp_path):
    async def bad(texts):
        return [[1.0, 0.1, 0.0] for _ in texts]   # |v| = 1.005: out of spec (tests/test_kb.py:1851-1875)

    kb = svs_amd.KB(str(tmp_path / "c.sqlite"), bad, index_factory=OracleIndex)
    with pytest.raises(ValueError, match="embedding magnitude out of spec"):
        with kb.bulk_add_docs() as add_doc:
            add_doc("first doc")
    assert len(kb) == 0   # the transaction rolled back
    kb.close()


def test_empty_kb_retrieve_raises_like_numpy(tmp_path):
    cases = load_cases()
    kb = svs_amd.KB(str(tmp_path / "d.sqlite"), embedding_func_from(cases), index_factory=OracleIndex)
    with pytest.raises(ValueError):   # (0,0) matrix: numpy "shapes ... not aligned"
        kb.retrieve("... first ...", 1)
    kb.close()


def test_invalidate_while_search_in_flight():
    """A released matrix must not break a search that already holds the index
    (reference: the closure keeps the arrays alive, kb.py:1180-1190)."""
    m = np.eye(4, dtype=np.float32)
    cache = mx.DeviceEmbeddingsMatrix(builder=lambda db: (m, np.array([10, 20, 30, 40])), index_factory=OracleIndex)
    cache.get_sync(None)
    idx, lookup = cache.hold()
    cache.invalidate()
    assert [int(lookup.arr[r]) for _, r in idx.search(m[2], 1)] == [30]
    idx.release()
    assert OracleIndex.live == 0
    with pytest.raises(RuntimeError):
        cache.search(m[0], 1)


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/svs"), reason="reference not present (GPU box)")
def test_attach_to_the_real_reference_kb(tmp_path):
    """attach(): the reference's OWN retrieve() code runs unchanged; its np.dot /
    np.argpartition calls are served through __array_function__."""
    sys.path.insert(0, "/root/reference/src")
    try:
        import svs
        cases = load_cases()
        kb = svs.KB(str(tmp_path / "ref.sqlite"), embedding_func_from(cases))
        svs_amd.attach(kb, index_factory=OracleIndex)
        assert isinstance(kb.embeddings_matrix, mx.DeviceEmbeddingsMatrix)
        replay_sync(kb, cases)
        # pairwise path still works through the host copy
        with kb.bulk_add_docs() as add_doc:
            add_doc("first doc"); add_doc("second doc"); add_doc("third doc")
        assert len(kb.document_top_pairwise_scores(2)) == 2
        kb.close()
        assert OracleIndex.live == 0
    finally:
        sys.path.remove("/root/reference/src")
        for k in [k for k in sys.modules if k == "svs" or k.startswith("svs.")]:
            del sys.modules[k]


def test_embedding_blocks_equal_the_matrix_build(tmp_path):
    """Streaming cold start: the (ids, rows) blocks are exactly the rows of
    build_embeddings_matrix (reference src/svs/kb.py:573-618; KATs tests/test_kb.py:753-806), whatever the
    block size; the empty table gives (0, 0) and no block."""
    from svs_amd.kb import _Store, embedding_to_bytes
    st = _Store(str(tmp_path / "blocks.sqlite"))
    with st.transaction():
        assert list(st.embedding_blocks()) == [(0, 0)]
    rng = np.random.default_rng(0)
    vecs = rng.standard_normal((1000, 12)).astype(np.float32)
    with st.transaction():
        for i, v in enumerate(vecs):
            d = st.add_doc(f"doc {i}", None, None)
            st.set_doc_embedding(d, embedding_to_bytes([float(x) for x in v]))
        st.del_doc(4); st.del_doc(777)                      # ids are not contiguous after deletes
    with st.transaction():
        m, lk = st.build_embeddings_matrix()
    for block_bytes in (48, 12 * 4 * 7, 1 << 20):
        with st.transaction():
            it = st.embedding_blocks(block_bytes)
            assert next(it) == m.shape
            ids, rows = [], []
            for a, b in it:
                assert len(a) * 12 * 4 <= max(block_bytes, 48)
                ids.append(a.copy()); rows.append(b.copy())   # the block buffer is reused
        assert np.array_equal(np.concatenate(ids), lk) and np.array_equal(np.vstack(rows), m)
    st.close()
