"""
SearchCoalescer: concurrent single-query searches share corpus passes.

``AsyncKB.retrieve`` runs its search on an executor thread outside the KB lock
(reference src/svs/kb.py:1184-1190), so a server with many tasks in flight has many threads
inside ``np.dot`` at once -- each streaming the whole matrix for itself.  On the GPU one pass over
the corpus serves 16 queries in the time it serves one (the score stage is HBM-bound:
1.0 ms for 16 queries vs 0.85 ms for 1 at 1M x 1536 f32), so searches that arrive
while another is in flight are queued and go out TOGETHER as one ``search_batch`` when the device is
free.  Nothing waits for company: a search that finds the device idle runs at once, alone, through
the single-query kernels exactly as without this class.

Results: the same total order (score desc, row desc) applied to scores that can differ from the solo
kernels' in the last bits (the batched kernels sum in another order, far inside the 1e-5 the
reference's own BLAS leaves open); rows closer than that rounding noise could swap.  On every golden
corpus recorded from the reference the coalesced rows equal the reference's position by position
(tests/test_search_gpu.py::test_search_golden).
"""
from __future__ import annotations

import threading
from typing import Any, List, Optional, Tuple

import numpy as np


def _as_list(res) -> List[Tuple[float, int]]:
    """(scores f32, rows i64) -> [(float, int)] as ``DeviceIndex.search`` returns it (float(np.float32) is the
    exact widening the reference's ``float(scores[i])`` does, src/svs/util.py:203)."""
    if isinstance(res, list):
        return res
    s, r = res
    return list(zip(s.astype(np.float64).tolist(), r.tolist()))


class _Req:
    __slots__ = ("q", "n", "event", "lead", "result", "error")

    def __init__(self, q: np.ndarray, n: int):
        self.q, self.n = q, n
        self.event = threading.Event()
        self.lead = False
        self.result: Any = None
        self.error: Optional[BaseException] = None


class SearchCoalescer:
    """One per loaded index generation (``DeviceEmbeddingsMatrix`` makes a new one with every
    index it builds, so queued requests never mix corpora)."""

    def __init__(self, max_batch: int = 256):
        self.max_batch = int(max_batch)
        self._mu = threading.Lock()
        self._pending: List[_Req] = []
        self._busy = False
        self.batches = 0        # statistics: corpus passes made ...
        self.queries = 0        # ... for this many searches

    def search(self, index: Any, query_vec: np.ndarray, n: int) -> List[Tuple[float, int]]:
        """``index.search(query_vec, n)``, possibly in the company of other callers' queries.
        ``index`` is the caller's own reference to the generation's index."""
        assert isinstance(n, int)
        q = np.asarray(query_vec, dtype=np.float32)
        if n <= 0 or n > 2048 or q.ndim != 1 or q.shape[0] != getattr(index, "d", -1) or not hasattr(index, "search_batch"):
            return index.search(query_vec, n)          # (errors and empty answers stay the caller's own)
        req = _Req(q, n)
        with self._mu:
            self._pending.append(req)
            if not self._busy:
                self._busy = True
                req.lead = True
        if not req.lead:
            req.event.wait()
            if not req.lead:                            # served by somebody else's pass
                if req.error is not None:
                    raise req.error
                return _as_list(req.result)
        # this thread drives the device until its own request is answered, then hands over
        while True:
            with self._mu:
                batch = self._pending[: self.max_batch]
                del self._pending[: len(batch)]
            self._run(index, batch)
            if req.result is not None or req.error is not None:
                break
        with self._mu:
            if self._pending:
                nxt = self._pending[0]                  # stays queued: its own loop takes it out
                nxt.lead = True
                nxt.event.set()
            else:
                self._busy = False
        if req.error is not None:
            raise req.error
        return _as_list(req.result)

    def _run(self, index: Any, batch: List[_Req]) -> None:
        try:
            if len(batch) == 1:
                batch[0].result = index.search(batch[0].q, batch[0].n)
            else:
                kmax = max(r.n for r in batch)
                s, rows = index.search_batch(np.stack([r.q for r in batch]), kmax)
                for i, r in enumerate(batch):
                    c = min(r.n, s.shape[1])            # a top-k list's first n entries ARE the top-n list
                    r.result = (s[i, :c], rows[i, :c])  # (each caller turns its own row into Python objects:
                                                        #  the thread that drives the device hands over first)
            self.batches += 1
            self.queries += len(batch)
        except BaseException as e:                      # noqa: BLE001 -- every waiter must be released
            for r in batch:
                if r.result is None:
                    r.error = e
        finally:
            for r in batch:
                if not r.lead:
                    r.event.set()
