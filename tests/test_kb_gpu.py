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
