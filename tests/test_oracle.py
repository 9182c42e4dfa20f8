"""CPU: the numpy restatement (oracle/) against the golden vectors recorded from
the real reference by tests/golden/make_golden.py."""
import json
import os
import sqlite3

import numpy as np
import pytest

from oracle import svs_oracle as oracle
from synth import corpus_and_query

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def test_top_k_cases_match_reference():
    g = _load("topk_cases.json")
    assert len(g["cases"]) >= 90
    for c in g["cases"]:
        arr = np.array(c["scores"], dtype=c["dtype"])
        got = oracle.cpu_top_k(arr, c["k"])
        assert [[s, i] for s, i in got] == c["expected"], c
        # where defined, the rule-based total order is the same thing
        assert oracle.total_order_top_k(arr, c["k"]) == got


def test_top_k_boundary_ties_scores_only():
    for c in _load("topk_cases.json")["boundary_ties"]:
        arr = np.array(c["scores"], dtype=c["dtype"])
        assert [s for s, _ in oracle.cpu_top_k(arr, c["k"])] == c["expected_scores"]
        assert [s for s, _ in oracle.total_order_top_k(arr, c["k"])] == c["expected_scores"]


def test_top_k_asserts_like_reference():
    with pytest.raises(AssertionError):
        oracle.cpu_top_k(np.zeros((2, 2), dtype=np.float32), 1)
    with pytest.raises(AssertionError):
        oracle.cpu_top_k(np.zeros(4, dtype=np.float32), np.int64(2))


@pytest.mark.parametrize("case", [c for c in _load("search_cases.json")["cases"] if c["n"] * c["d"] <= 110_000_000],
                         ids=lambda c: f'{c["kind"]}-{c["n"]}x{c["d"]}-k{c["k"]}')
def test_search_cases_match_reference(case):
    m, qs = corpus_and_query(case["kind"], case["seed"], case["n"], case["d"], case["nq"])
    for qi, q in enumerate(qs):
        top = oracle.cpu_search(m, q, case["k"])
        assert [i for _, i in top] == case["rows"][qi]
        assert [s for s, _ in top] == case["scores"][qi]


def test_codec_and_matrix_build():
    g = _load("kb_cases.json")
    for c in g["codec"]:
        assert oracle.embedding_to_bytes(c["values"]).hex() == c["hex"]
        got = oracle.embedding_from_bytes(bytes.fromhex(c["hex"]))
        assert np.allclose(got, np.array(c["values"], dtype=np.float32))
    conn = sqlite3.connect(":memory:")
    conn.execute("CREATE TABLE embeddings (id INTEGER PRIMARY KEY, embedding BLOB NOT NULL) STRICT;")
    step0, step1 = g["matrix_build"]
    for b in step0["blobs_hex"]:
        conn.execute("INSERT INTO embeddings (embedding) VALUES (?);", (bytes.fromhex(b),))
    m, lk = oracle.build_embeddings_matrix(conn)
    assert m.dtype == np.float32 and lk.dtype == np.int64
    assert m.tolist() == step0["matrix"] and lk.tolist() == step0["lookup"]
    conn.execute("DELETE FROM embeddings WHERE id = ?;", (step1["deleted_doc"],))
    m, lk = oracle.build_embeddings_matrix(conn)
    assert m.tolist() == step1["matrix"] and lk.tolist() == step1["lookup"]
    conn.execute("DELETE FROM embeddings;")
    m, lk = oracle.build_embeddings_matrix(conn)
    assert m.shape == (0, 0) and lk.shape == (0,)


def test_top_pairs():
    g = _load("kb_cases.json")["top_pairs"]
    got = oracle.cpu_top_pairs(np.array(g["matrix"], dtype=np.float32), g["k"])
    assert [list(p) for p in got] == g["expected"]


def test_magnitude_guard():
    oracle.check_magnitude([[1.0, 0.001, 0.0], [0.707, 0.707, 0.0]])
    with pytest.raises(ValueError):
        oracle.check_magnitude([[1.0, 0.1, 0.0]])
