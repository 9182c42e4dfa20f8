"""GPU: the reference's retrieve() transcript (tests/golden/kb_cases.json,
captured from the real svs.KB) replayed on the HIP path through the KB mirror."""
import asyncio

import numpy as np
import pytest

from kb_transcript import embedding_func_from, load_cases, replay_async, replay_sync

pytestmark = pytest.mark.gpu


def test_sync_kb_transcript_hip(gpu, tmp_path):
    import svs_amd
    cases = load_cases()
    kb = svs_amd.KB(str(tmp_path / "a.sqlite"), embedding_func_from(cases))
    replay_sync(kb, cases)
    assert isinstance(kb.embeddings_matrix.index, svs_amd.DeviceIndex)
    kb.close()


def test_async_kb_transcript_hip(gpu, tmp_path):
    import svs_amd
    cases = load_cases()

    async def run():
        kb = svs_amd.AsyncKB(str(tmp_path / "b.sqlite"), embedding_func_from(cases))
        await replay_async(kb, cases)
        await kb.close()

    asyncio.run(run())


def test_async_concurrent_retrieves_and_invalidate(gpu, tmp_path):
    """Many retrieve() coroutines at once (superheavy runs on executor threads,
    outside the lock) with an add in the middle."""
    import svs_amd
    rng = np.random.default_rng(0)
    vecs = rng.standard_normal((400, 64))
    vecs /= np.linalg.norm(vecs, axis=1, keepdims=True)
    table = {f"doc {i}": [float(x) for x in vecs[i]] for i in range(400)}

    async def ef(texts):
        return [table[t] for t in texts]

    async def run():
        kb = svs_amd.AsyncKB(str(tmp_path / "c.sqlite"), ef)
        async with kb.bulk_add_docs() as add_doc:
            for i in range(300):
                await add_doc(f"doc {i}")

        async def one(i):
            docs = await kb.retrieve(f"doc {i}", 3)
            assert docs[0]["doc"]["text"] == f"doc {i}" and abs(docs[0]["score"] - 1.0) < 1e-5

        async def adder():
            async with kb.bulk_add_docs() as add_doc:
                for i in range(300, 400):
                    await add_doc(f"doc {i}")

        await asyncio.gather(*[one(i) for i in range(0, 300, 7)], adder(), *[one(i) for i in range(3, 300, 11)])
        docs = await kb.retrieve("doc 399", 1)
        assert docs[0]["doc"]["text"] == "doc 399"
        await kb.close()

    asyncio.run(run())


@pytest.mark.parametrize("dtype", ["f16", "fp8"])
def test_kb_with_a_reduced_precision_corpus(gpu, tmp_path, dtype):
    """KB(..., dtype=...) / attach(kb, dtype=...): same retrieve() surface, corpus stored as
    halves or e4m3 in HBM.  Every document must still find itself first."""
    import svs_amd
    rng = np.random.default_rng(3)
    vecs = rng.standard_normal((500, 384))
    vecs /= np.linalg.norm(vecs, axis=1, keepdims=True)
    table = {f"doc {i}": [float(x) for x in vecs[i]] for i in range(500)}

    async def ef(texts):
        return [table[t] for t in texts]

    kb = svs_amd.KB(str(tmp_path / f"{dtype}.sqlite"), ef, dtype=dtype)
    with kb.bulk_add_docs() as add_doc:
        for i in range(500):
            add_doc(f"doc {i}")
    for i in (0, 17, 499):
        docs = kb.retrieve(f"doc {i}", 5)
        assert docs[0]["doc"]["text"] == f"doc {i}" and abs(docs[0]["score"] - 1.0) < (2e-3 if dtype == "f16" else 3e-2)
        assert len(docs) == 5 and all(docs[j]["score"] >= docs[j + 1]["score"] for j in range(4))
    assert kb.embeddings_matrix.index.dtype == dtype
    many = kb.retrieve_many([f"doc {i}" for i in range(40)], 3)
    assert [r[0]["doc"]["text"] for r in many] == [f"doc {i}" for i in range(40)]
    kb.close()


def test_async_kb_retrieve_many_and_pairs_on_hip(gpu, tmp_path):
    """AsyncKB.retrieve_many / document_top_pairwise_scores (reference src/svs/kb.py:1208-1243) on the
    HIP path: equal to the sync KB over the same file; element i of retrieve_many equals retrieve(i)."""
    import asyncio
    import svs_amd
    rng = np.random.default_rng(11)
    vecs = rng.standard_normal((600, 256))
    vecs /= np.linalg.norm(vecs, axis=1, keepdims=True)
    table = {f"doc {i}": [float(x) for x in vecs[i]] for i in range(600)}

    async def ef(texts):
        return [table[t] for t in texts]

    path = str(tmp_path / "async_many.sqlite")
    kb = svs_amd.KB(path, ef)
    with kb.bulk_add_docs() as add_doc:
        for i in range(500):
            add_doc(f"doc {i}")
    qs = [f"doc {i}" for i in list(range(0, 500, 13)) + [550, 599]]
    want_many = kb.retrieve_many(qs, 9)
    want_pairs = kb.document_top_pairwise_scores(50)
    kb.close()

    async def run():
        akb = svs_amd.AsyncKB(path, ef)
        got_many, got_pairs = await asyncio.gather(akb.retrieve_many(qs, 9), akb.document_top_pairwise_scores(50))
        assert got_many == want_many
        assert [(s, a["id"], b["id"]) for s, a, b in got_pairs] == [(s, a["id"], b["id"]) for s, a, b in want_pairs]
        for i in (0, 5, len(qs) - 1):     # (single-query GEMV vs batched MFMA kernel: other summation order)
            one = await akb.retrieve(qs[i], 9)
            assert [d["doc"]["id"] for d in one] == [d["doc"]["id"] for d in got_many[i]]
            assert max(abs(a["score"] - b["score"]) for a, b in zip(one, got_many[i])) <= 1e-5
        await akb.close()

    asyncio.run(run())


@pytest.mark.parametrize("dtype,d", [("f32", 2048), ("f16", 2048), ("fp8", 2048), ("f32", 1000)])
def test_streaming_cold_start_equals_matrix_upload(gpu, tmp_path, dtype, d):
    """KB.load() decodes BLOBs straight into the library's pinned staging blocks and commits them
    (svs_index_staging_acquire / commit / finish: no (n, m) host matrix, one host copy per byte): the
    index must hold exactly what DeviceIndex(build_embeddings_matrix()) holds, ids included
    (non-contiguous after a delete).  d = 2048: a staging block holds 4,096 rows, so 4,500 rows take two
    blocks; d = 1000: rows padded to ld = 1,024 in HBM (the 2-D copy)."""
    import svs_amd
    from svs_amd import DeviceIndex
    n = 4500
    rng = np.random.default_rng(5)
    vecs = rng.standard_normal((n, d)).astype(np.float32)
    vecs /= np.linalg.norm(vecs, axis=1, keepdims=True)
    path = str(tmp_path / f"cold_{dtype}_{d}.sqlite")
    from svs_amd.kb import _Store
    st = _Store(path)                                  # written directly: 9M python floats through an embedding func would take minutes
    with st.transaction():
        st.conn.executemany("INSERT INTO embeddings (embedding) VALUES (?)", [(r.tobytes(),) for r in vecs])
        st.conn.executemany("INSERT INTO docs (parent_id, level, text, embedding, meta) VALUES (NULL, 0, ?, ?, NULL)",
                            [(f"doc {i}", i + 1) for i in range(n)])
        st.del_doc(7); st.del_doc(n)
    st.close()

    async def ef(texts):
        return [[float(x) for x in vecs[int(t.split()[1])]] for t in texts]

    kb = svs_amd.KB(path, ef, dtype=dtype)             # cold
    kb.load()
    with kb.db.transaction():
        m, lk = kb.db.build_embeddings_matrix()
    assert m.shape == (n - 2, d)
    assert kb.embeddings_matrix.embeddings_matrix is None                # no host copy kept
    assert np.array_equal(kb.embeddings_matrix.emb_id_lookup, lk) and 7 not in lk and 1 in lk
    ref = DeviceIndex(m, dtype=dtype)
    idx = kb.embeddings_matrix.index
    assert idx.shape == ref.shape
    assert np.array_equal(idx.stored_rows(), ref.stored_rows())
    if dtype == "fp8":
        q = vecs[1234]
        assert idx.search(q, 10) == ref.search(q, 10)                    # row scales too
    docs = kb.retrieve("doc 1234", 5)
    assert docs[0]["doc"]["text"] == "doc 1234"
    # a search issued right after the last commit waits for the pending copies itself
    idx2 = DeviceIndex.empty(d, dtype=dtype, reserve=100)
    blk = idx2.staging_acquire()
    blk[:50] = vecs[:50]
    idx2.staging_commit(50)
    assert idx2.search(vecs[3], 1)[0][1] == 3
    idx2.staging_finish()
    assert idx2.shape == (50, d)
    idx2.release()
    # ... and so does every other reader of the rows (scores, pairwise, the debug read-back): a full block this
    # time (32 MiB in flight behind the commit), each reader on a fresh index, before svs_index_staging_finish
    full = (32 << 20) // (d * 4)
    big = np.tile(vecs, (full // n + 1, 1))[:full]
    want = DeviceIndex(big[:full], dtype=dtype)
    for reader in ("scores", "stored_rows", "top_pairs"):
        idx3 = DeviceIndex.empty(d, dtype=dtype, reserve=full)
        blk = idx3.staging_acquire()
        blk[:full] = big
        idx3.staging_commit(full)
        if reader == "scores":
            assert np.array_equal(idx3.scores(vecs[11]), want.scores(vecs[11]))
        elif reader == "stored_rows":
            assert np.array_equal(idx3.stored_rows(full - 64, 64), want.stored_rows(full - 64, 64))
        else:
            assert idx3.top_pairs(5) == want.top_pairs(5)
        idx3.staging_finish()
        idx3.release()
    want.release(); ref.release(); kb.close()
