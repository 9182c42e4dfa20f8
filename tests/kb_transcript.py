"""Replays the retrieve() transcript captured from the reference
(tests/golden/kb_cases.json; it follows reference tests/test_kb.py:1755-1846)
against any object with the KB surface."""
import json
import os

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_cases():
    with open(os.path.join(GOLD, "kb_cases.json")) as f:
        return json.load(f)


def embedding_func_from(cases):
    vecs = cases["embedding_map"]

    async def embedding_func(texts):
        out = []
        for t in texts:
            for key, v in vecs.items():
                if key in t:
                    out.append(v)
                    break
            else:
                raise ValueError("unexpected doc")
        return out

    return embedding_func


def _check(step, docs):
    assert [d["doc"]["text"] for d in docs] == step["texts"], step
    assert [d["doc"]["id"] for d in docs] == step["ids"], step
    assert len(docs) == len(step["scores"])
    for d, s in zip(docs, step["scores"]):
        assert isinstance(d["score"], float)
        assert abs(d["score"] - s) <= 1e-5, (d["score"], s)
        assert set(d["doc"]) == {"id", "parent_id", "level", "text", "embedding", "meta"}


def replay_sync(kb, cases):
    for step in cases["script"]:
        if step["op"] == "add":
            with kb.bulk_add_docs() as add_doc:
                assert add_doc(step["text"]) == step["id"]
        elif step["op"] == "del":
            with kb.bulk_del_docs() as del_doc:
                del_doc(step["id"])
        else:
            _check(step, kb.retrieve(step["query"], n=step["n"]))


async def replay_async(kb, cases):
    for step in cases["script"]:
        if step["op"] == "add":
            async with kb.bulk_add_docs() as add_doc:
                assert await add_doc(step["text"]) == step["id"]
        elif step["op"] == "del":
            async with kb.bulk_del_docs() as del_doc:
                await del_doc(step["id"])
        else:
            _check(step, await kb.retrieve(step["query"], n=step["n"]))
