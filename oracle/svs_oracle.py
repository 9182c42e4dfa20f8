"""
CPU oracle for the svs brute-force similarity path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain numpy, the algorithm the reference (Rhobota/svs
v0.7.4) runs inside ``KB.retrieve()`` / ``AsyncKB.retrieve()``.  It is the
checker for the HIP path, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  Nothing under ``svs_amd/`` imports this module.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imported the real
reference in the build container and recorded its outputs on (i) every
``get_top_k`` case of the reference's ``tests/test_util.py:142-400``, (ii) the
tie table of SURVEY.md section 8(c), (iii) seeded synthetic corpora up to
1M x 1536.  ``tests/test_oracle.py`` checks this restatement against those
fixtures; the script also asserted equality with the live reference when it
was run.

Each function cites the reference file:line (relative to the reference repo)
that it follows.
"""
from __future__ import annotations

import struct
from typing import List, Optional, Sequence, Tuple

import numpy as np

# src/svs/kb.py:58
EMBEDDING_MAGNITUDE_TOLERANCE = 0.001


# --------------------------------------------------------------------------
# A3: scores = np.dot(embeddings_matrix, query_vec)         src/svs/kb.py:1623
# --------------------------------------------------------------------------
def cpu_scores(embeddings_matrix: np.ndarray, query_vec: np.ndarray) -> np.ndarray:
    """f32 GEMV exactly as the reference issues it (numpy -> BLAS sgemv).

    src/svs/kb.py:1623 (sync) and :1185 (async): ``x = np.dot(M, q)``.
    A shape mismatch raises numpy's ValueError, as in the reference.
    """
    return np.dot(embeddings_matrix, query_vec)


def cpu_scores_f64(embeddings_matrix: np.ndarray, query_vec: np.ndarray) -> np.ndarray:
    """Same dot products accumulated in f64 (the "truth" used to bound f32
    accumulation noise when two implementations disagree on a near tie)."""
    return np.dot(embeddings_matrix.astype(np.float64), query_vec.astype(np.float64))


# --------------------------------------------------------------------------
# A4: get_top_k                                         src/svs/util.py:190-203
# --------------------------------------------------------------------------
def cpu_top_k(scores: np.ndarray, top_k: int) -> List[Tuple[float, int]]:
    """Restatement of ``svs.util.get_top_k`` (src/svs/util.py:190-203).

    asserts (``:196-197``), clamp k to len (``:198-199``), k<=0 -> []
    (``:200-201``), argpartition (``:202``), sort of (score, index) tuples
    reversed (``:203``) -> order is score desc, ties index desc.
    """
    assert scores.ndim == 1
    assert isinstance(top_k, int)
    if top_k > len(scores):
        top_k = len(scores)
    if top_k <= 0:
        return []
    indices = np.argpartition(scores, -top_k)[-top_k:]
    return sorted([(float(scores[i]), int(i)) for i in indices], reverse=True)


def total_order_top_k(scores: np.ndarray, top_k: int) -> List[Tuple[float, int]]:
    """Rule-based top-k under the total order (score desc, index desc).

    Equal to ``cpu_top_k`` whenever the k-th and (k+1)-th scores differ
    (SURVEY.md section 8(a) row A4); at a boundary tie the reference's choice
    is introselect-internal, and this rule (largest index wins) is the one the
    HIP path implements.
    """
    n = len(scores)
    k = max(0, min(int(top_k), n))
    if k == 0:
        return []
    idx = np.arange(n, dtype=np.int64)
    order = np.lexsort((idx, scores))[::-1][:k]  # primary: score, secondary: index
    return [(float(scores[i]), int(i)) for i in order]


# --------------------------------------------------------------------------
# A2: superheavy()                                     src/svs/kb.py:1622-1627
# --------------------------------------------------------------------------
def cpu_search(
    embeddings_matrix: np.ndarray,
    query_vec: np.ndarray,
    n: int,
    emb_id_lookup: Optional[np.ndarray] = None,
) -> List[Tuple[float, int]]:
    """``superheavy()`` of src/svs/kb.py:1622-1627 / :1184-1189.

    With ``emb_id_lookup`` the second tuple member is the embedding id
    (``int(emb_id_lookup[index])``, kb.py:1626); without, the row index.
    """
    x = cpu_scores(embeddings_matrix, query_vec)
    out = []
    for score, index in cpu_top_k(x, n):
        out.append((score, int(emb_id_lookup[index]) if emb_id_lookup is not None else index))
    return out


def cpu_search_batch(
    embeddings_matrix: np.ndarray, queries: np.ndarray, n: int
) -> List[List[Tuple[float, int]]]:
    """Loop of A2 over the rows of ``queries`` (the reference has no batched
    entry; a batch is by definition the per-query results in order)."""
    return [cpu_search(embeddings_matrix, q, n) for q in queries]


# --------------------------------------------------------------------------
# pairwise (next row f2)                     src/svs/util.py:206-233, kb.py:1651
# --------------------------------------------------------------------------
def cpu_top_pairs(pairwise: np.ndarray, top_k: int) -> List[Tuple[float, int, int]]:
    """``svs.util.get_top_pairs`` (src/svs/util.py:206-233): strict upper
    triangle, flattened row-major, then ``get_top_k``."""
    assert len(pairwise.shape) == 2
    rows, cols = pairwise.shape
    assert rows == cols
    indices = np.triu_indices_from(pairwise, k=1)
    vals = pairwise[indices]
    top = cpu_top_k(vals, top_k)
    return [(s, int(indices[0][ii]), int(indices[1][ii])) for s, ii in top]


# --------------------------------------------------------------------------
# A7/A8: codec, matrix build, magnitude guard
# --------------------------------------------------------------------------
def embedding_to_bytes(embedding: Sequence[float]) -> bytes:
    """src/svs/embeddings/util.py:15-16 -- little-endian f32 BLOB."""
    return struct.pack(f"<{len(embedding)}f", *embedding)


def embedding_from_bytes(blob: bytes) -> List[float]:
    """src/svs/embeddings/util.py:19-23."""
    size = struct.calcsize("<f")
    assert (len(blob) % size) == 0
    return list(struct.unpack(f"<{len(blob) // size}f", blob))


def build_embeddings_matrix(conn) -> Tuple[np.ndarray, np.ndarray]:
    """``_Querier.build_embeddings_matrix`` (src/svs/kb.py:573-618): rows in
    ``SELECT id, embedding FROM embeddings`` order; m from the first row;
    empty table -> shape (0, 0)."""
    n = conn.execute("SELECT COUNT(*) FROM embeddings;").fetchone()[0]
    row = conn.execute("SELECT embedding FROM embeddings LIMIT 1;").fetchone()
    m = len(embedding_from_bytes(row[0])) if row is not None else 0
    matrix = np.zeros((n, m), dtype=np.float32)
    lookup = np.zeros(n, dtype=np.int64)
    i = -1
    for i, r in enumerate(conn.execute("SELECT id, embedding FROM embeddings;")):
        e = embedding_from_bytes(r[1])
        assert len(e) == m
        matrix[i] = e
        lookup[i] = r[0]
    assert i == n - 1
    return matrix, lookup


def check_magnitude(vectors: Sequence[Sequence[float]], tolerance: float = EMBEDDING_MAGNITUDE_TOLERANCE) -> None:
    """``wrap_embeddings_func_check_magnitude`` body
    (src/svs/embeddings/util.py:34-39): f32 norms, raise ValueError outside
    1 +/- tolerance."""
    v = np.array(vectors, dtype=np.float32)
    mags = np.sqrt((v * v).sum(axis=1))
    if (np.abs(mags - 1.0) > tolerance).any():
        raise ValueError("embedding magnitude out of spec")


def query_vec_from_list(values: Sequence[float]) -> np.ndarray:
    """src/svs/kb.py:1620 -- python floats -> f32 round-to-nearest."""
    return np.array(values, dtype=np.float32)
