"""
Seeded synthetic corpora.  Shared by the golden-vector
generator, the parity tests and bench.py's CPU leg so that every consumer
rebuilds bit-identical f32 inputs from (kind, seed, n, d).

* ``gaussian``: primary, parity-gated corpus.  standard_normal in f64,
  row-normalised in f64, cast to f32.  Adjacent top-100 score gaps are ~40x the
  f32 accumulation noise, so the ordered index list is well defined.
* ``uniform``: the reference's own benchmark recipe
  (examples/One Million Documents Benchmark.ipynb cell 5:
  ``np.random.random((n, 1536))`` then row-normalise).  Scores cluster in
  [0.71, 0.78]; used as a near-tie stress, compared by score.

The query is drawn from the same generator *after* the corpus.
"""
from __future__ import annotations

import numpy as np

BLOCK_ROWS = 50_000


def _draw(rng: np.random.Generator, kind: str, rows: int, d: int) -> np.ndarray:
    if kind == "gaussian":
        x = rng.standard_normal((rows, d))
    elif kind == "uniform":
        x = rng.random((rows, d))
    else:
        raise ValueError(kind)
    x /= np.sqrt((x * x).sum(axis=1, keepdims=True))
    return x.astype(np.float32)


def corpus_and_query(kind: str, seed: int, n: int, d: int, nq: int = 1):
    """Returns (M f32 (n,d) C-contiguous, Q f32 (nq,d))."""
    rng = np.random.default_rng(seed)
    m = np.empty((n, d), dtype=np.float32)
    for r0 in range(0, n, BLOCK_ROWS):
        r1 = min(n, r0 + BLOCK_ROWS)
        m[r0:r1] = _draw(rng, kind, r1 - r0, d)
    q = _draw(rng, kind, nq, d)
    return m, q
