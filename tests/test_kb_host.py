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
    assert [int(lookup[r]) for _, r in idx.search(m[2], 1)] == [30]
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
