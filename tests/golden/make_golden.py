#!/usr/bin/env python3
"""
Generates the golden fixtures in this directory by running the REAL reference
(Rhobota/svs v0.7.4, imported from /root/reference/src) in the build container.

    python3 tests/golden/make_golden.py [--skip-1m]

The reference cannot travel to the GPU box, so only its *outputs* are kept:
  topk_cases.json    inputs/outputs of svs.util.get_top_k (A4)
  search_cases.json  per seeded corpus: ordered rows + f32 scores of
                     np.dot + get_top_k exactly as kb.py:1622-1627 runs them
  kb_cases.json      captured KB.retrieve() transcripts + matrix-build KATs
While generating, every output is also compared with oracle/svs_oracle.py
(the restatement); a mismatch aborts.
"""
import argparse
import asyncio
import itertools
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference/src")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import svs  # noqa: E402  (the reference)
from svs.util import get_top_k, get_top_pairs  # noqa: E402
from svs.embeddings.util import embedding_to_bytes, embedding_from_bytes  # noqa: E402
from svs.kb import _DB  # noqa: E402

from oracle import svs_oracle as oracle  # noqa: E402
from synth import corpus_and_query  # noqa: E402


def topk_cases():
    cases = []

    def add(scores, dtype, k, note):
        arr = np.array(scores, dtype=dtype)
        got = get_top_k(arr, k)
        assert got == oracle.cpu_top_k(arr, k), (scores, k)
        cases.append({
            "note": note, "dtype": dtype, "scores": [float(x) for x in arr], "k": k,
            "expected": [[s, i] for s, i in got],
        })

    # every array the reference's tests/test_util.py:142-400 uses: [], 1, 2 and
    # all permutations of three distinct values, each with k = 0..len+1
    base = [[], [0.4], [0.4, 0.2], [0.2, 0.4]] + [list(p) for p in itertools.permutations([0.2, 0.4, 0.8])]
    for arr in base:
        for k in range(0, len(arr) + 2):
            add(arr, "float64", k, "tests/test_util.py:142-400")
            add(arr, "float32", k, "same inputs as f32")
    # SURVEY.md 8(c) tie table (in-set ties are ordered index DESC)
    tie = [.5, .9, .5, .9, .5, .1]
    add(tie, "float32", 2, "tie table k=2")
    add(tie, "float32", 6, "tie table k=6 (full)")
    add(tie, "float32", 5, "tie table k=5 (boundary is 0.5 > 0.1: defined)")
    # negative / mixed-sign / zero scores
    add([-0.5, 0.25, -0.75, 0.0, 0.125], "float32", 3, "mixed sign")
    add([-1.0, -2.0, -3.0], "float32", 2, "all negative")
    # seeded random cases, distinct values
    rng = np.random.default_rng(99)
    for n, k in [(17, 5), (64, 64), (65, 1), (300, 100), (1000, 7), (4096, 128), (5000, 1000)]:
        add(rng.standard_normal(n).astype(np.float32).tolist(), "float32", k, f"random n={n}")
    return cases


def boundary_tie_cases():
    """Cases whose k-th and (k+1)-th scores are equal: the reference's pick is
    introselect-internal, so only its SCORES are pinned."""
    out = []
    tie = np.array([.5, .9, .5, .9, .5, .1], dtype=np.float32)
    for k in (1, 3, 4):
        got = get_top_k(tie, k)
        out.append({"dtype": "float32", "scores": [float(x) for x in tie], "k": k,
                    "expected_scores": [s for s, _ in got]})
    eq = np.full(1000, 0.25, dtype=np.float32)
    got = get_top_k(eq, 5)
    out.append({"dtype": "float32", "scores": [float(x) for x in eq], "k": 5,
                "expected_scores": [s for s, _ in got]})
    return out


SEARCH_SPECS = [
    # (kind, seed, n, d, k, nq, note)
    ("gaussian", 1234, 10548, 1536, 100, 4, "cfg1 BASELINE.json configs[0]"),
    ("uniform", 1234, 10548, 1536, 100, 2, "cfg1, reference notebook recipe"),
    ("gaussian", 7, 65536, 1536, 100, 4, "65,536-row slice"),
    ("gaussian", 11, 4097, 100, 10, 2, "d % 4 == 0, not a multiple of 256"),
    ("gaussian", 12, 1000, 3, 5, 3, "d = 3 (reference unit tests' dim)"),
    ("gaussian", 13, 5000, 1537, 100, 2, "odd d"),
    ("gaussian", 14, 300, 768, 300, 2, "k == n, full ranking"),
    ("gaussian", 15, 20000, 3072, 100, 2, "d = 3072 (text-embedding-3-large)"),
    ("gaussian", 16, 100000, 256, 1000, 2, "large k"),
    ("gaussian", 17, 77, 64, 100, 2, "k > n clamps"),
    ("gaussian", 18, 1, 1536, 100, 1, "single row"),
    ("uniform", 21, 200000, 1536, 100, 2, "near-tie stress, compare by score"),
    ("gaussian", 1234, 1000000, 1536, 100, 4, "cfg2 BASELINE.json configs[1]"),
]


def search_cases(skip_1m: bool):
    out = []
    for kind, seed, n, d, k, nq, note in SEARCH_SPECS:
        if skip_1m and n >= 1000000:
            continue
        m, qs = corpus_and_query(kind, seed, n, d, nq)
        rows, scores, gaps = [], [], []
        for q in qs:
            # exactly superheavy(): kb.py:1623 + :1625
            x = np.dot(m, q)
            top = get_top_k(x, k)
            assert top == oracle.cpu_search(m, q, k), (kind, seed, n, d)
            rows.append([i for _, i in top])
            scores.append([s for s, _ in top])
            # margin: smallest adjacent gap among the top-(k+1) f64 scores
            x64 = oracle.cpu_scores_f64(m, q)
            kk = min(k + 1, n)
            best = np.sort(x64)[::-1][:kk]
            gaps.append(float(np.min(-np.diff(best))) if kk > 1 else None)
        out.append({"kind": kind, "seed": seed, "n": n, "d": d, "k": k, "nq": nq, "note": note,
                    "rows": rows, "scores": scores, "min_adjacent_gap_f64": gaps})
        print(f"  {kind} seed={seed} {n}x{d} k={k}: min gap {gaps}", flush=True)
        del m
    return out


def kb_cases():
    vecs = {
        "first": [1.0, 0.001, 0.0],
        "second": [0.0, 1.0, 0.0001],
        "third": [0.01, 0.0, 1.0],
        "forth": [0.707, 0.707, 0.0],
    }

    async def embedding_func(texts):
        ret = []
        for t in texts:
            for key, v in vecs.items():
                if key in t:
                    ret.append(v)
                    break
            else:
                raise ValueError("unexpected doc")
        return ret

    script = []
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "golden.sqlite")
        kb = svs.KB(path, embedding_func)

        def retrieve(query, n):
            docs = kb.retrieve(query, n=n)
            script.append({"op": "retrieve", "query": query, "n": n,
                           "texts": [d["doc"]["text"] for d in docs],
                           "ids": [d["doc"]["id"] for d in docs],
                           "scores": [d["score"] for d in docs]})

        def add(text):
            with kb.bulk_add_docs() as add_doc:
                script.append({"op": "add", "text": text, "id": add_doc(text)})

        def delete(doc_id):
            with kb.bulk_del_docs() as del_doc:
                del_doc(doc_id)
            script.append({"op": "del", "id": doc_id})

        # transcript follows tests/test_kb.py:1755-1846
        for t in ("third doc", "first doc", "second doc"):
            add(t)
        for qy in ("... first ...", "... second ...", "... third ..."):
            retrieve(qy, 3)
        retrieve("... forth ...", 1)
        add("forth doc")
        retrieve("... forth ...", 1)
        retrieve("... forth ...", 10)   # n > N clamps
        retrieve("... forth ...", 0)    # n == 0 -> []
        for i in (1, 2, 4):
            delete(i)
        retrieve("... forth ...", 1)
        retrieve("... first ...", 5)
        kb.close()

        # matrix-build KAT (tests/test_kb.py:753-806): BLOB rows -> arrays,
        # non-contiguous ids after a delete
        path2 = os.path.join(td, "matrix.sqlite")
        db = _DB(path2)
        steps = []
        with db as q:
            blobs = [b"\x00\x00\x80?\x00\x00`@", b"\x00\x00\x00@\x00\x00`@",
                     b"\x00\x00\x00@\x00\x00\x80?", b"\x00\x00`@\x00\x00\x80@"]
            for i, b in enumerate(blobs):
                q.add_doc(text=f"doc {i}", parent_id=None, meta=None, embedding=b)
            m, lk = q.build_embeddings_matrix()
            steps.append({"blobs_hex": [b.hex() for b in blobs], "matrix": m.tolist(), "lookup": lk.tolist()})
            q.del_doc(3)
            m, lk = q.build_embeddings_matrix()
            steps.append({"deleted_doc": 3, "matrix": m.tolist(), "lookup": lk.tolist()})
        db.close()

    codec = [{"values": v, "hex": embedding_to_bytes(v).hex()} for v in ([], [1.0], [1.0, 3.5], [0.1, -2.25, 3e-5])]
    for c in codec:
        assert oracle.embedding_to_bytes(c["values"]).hex() == c["hex"]
        assert oracle.embedding_from_bytes(bytes.fromhex(c["hex"])) == embedding_from_bytes(bytes.fromhex(c["hex"]))

    pair_m = np.array([[1, .2, .9, .4], [.2, 1, .3, .8], [.9, .3, 1, .5], [.4, .8, .5, 1]], dtype=np.float32)
    pairs = get_top_pairs(pair_m, 3)
    assert pairs == oracle.cpu_top_pairs(pair_m, 3)
    return {"embedding_map": vecs, "script": script, "matrix_build": steps, "codec": codec,
            "top_pairs": {"matrix": pair_m.tolist(), "k": 3, "expected": [list(p) for p in pairs]}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-1m", action="store_true")
    args = ap.parse_args()
    meta = {"reference": "Rhobota/svs", "svs_version": svs.__version__, "numpy": np.__version__}
    with open(os.path.join(HERE, "topk_cases.json"), "w") as f:
        json.dump({"meta": meta, "cases": topk_cases(), "boundary_ties": boundary_tie_cases()}, f)
    with open(os.path.join(HERE, "kb_cases.json"), "w") as f:
        json.dump({"meta": meta, **kb_cases()}, f)
    sc = search_cases(args.skip_1m)
    with open(os.path.join(HERE, "search_cases.json"), "w") as f:
        json.dump({"meta": meta, "cases": sc}, f)
    print("golden fixtures written")


if __name__ == "__main__":
    main()
