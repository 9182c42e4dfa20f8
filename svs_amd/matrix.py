"""
The seam between SVS's KB classes and the HIP backend.

``DeviceEmbeddingsMatrix`` has the surface of the reference's
``_EmbeddingsMatrix`` (src/svs/kb.py:856-893: ``get_sync`` / ``get`` /
``invalidate``) but what it caches is an HBM-resident ``DeviceIndex`` next to the
``emb_id_lookup`` array.  Two ways to put it under the reference's
``retrieve()``:

* the explicit 3-line patch of INTEGRATION.md (``superheavy()`` calls
  ``matrix.search(query_vec, n)``), or
* ``attach(kb)``: no reference line changes.  ``get_sync`` then returns a
  ``DeviceMatrixView`` in place of the numpy matrix; the reference's own
  ``np.dot(embeddings_matrix, query_vec)`` (kb.py:1623) and
  ``np.argpartition(scores, -top_k)`` (util.py:202) are routed to the device
  through NumPy's ``__array_function__`` protocol, and ``float(scores[i])``
  (util.py:203) reads the scores the device returned.  The reference's
  ``sorted(..., reverse=True)`` then orders them exactly as before.
"""
from __future__ import annotations

import asyncio
import logging
import threading
from typing import Any, Callable, List, Optional, Tuple

import numpy as np

from .coalesce import SearchCoalescer
from .index import DeviceIndex

_LOG = logging.getLogger(__name__)

MatrixBuilder = Callable[[Any], Tuple[np.ndarray, np.ndarray]]


def _default_builder(db) -> Tuple[np.ndarray, np.ndarray]:
    """``with db as q: q.build_embeddings_matrix()`` -- the reference's own
    producer (src/svs/kb.py:872-873)."""
    with db as q:
        return q.build_embeddings_matrix()


class _Lookup:
    """Row -> embedding-id table of one loaded matrix.  ``arr`` only ever grows
    (append) and masked rows are never returned, so a search that started before an
    append can map its rows through the CURRENT array; a search that outlives an
    ``invalidate()`` keeps this object (and so the right table) alive."""

    def __init__(self, arr: np.ndarray):
        self.arr = arr


class DeviceEmbeddingsMatrix:
    """Lazy cache of (DeviceIndex, emb_id_lookup); drop-in for
    ``svs.kb._EmbeddingsMatrix``."""

    # rebuild from storage instead of tombstoning once this share of the rows is dead
    COMPACT_AT = 0.25

    def __init__(self, device: int = 0, builder: MatrixBuilder = _default_builder,
                 index_factory: Callable[..., Any] = DeviceIndex, keep_host_matrix: bool = True,
                 view: bool = False, block_builder: Optional[Callable[[Any], Any]] = None):
        self.device = device
        self._builder = builder
        # optional streaming producer: db -> iterator of (n, m), then (ids, rows) blocks
        # (svs_amd.kb._blocks_from_store); used when no host copy is kept and the index type can
        # start empty and grow (DeviceIndex.empty / append)
        self._block_builder = block_builder
        self._index_factory = index_factory
        self._keep_host = keep_host_matrix
        self._view = view
        self._mu = threading.Lock()
        self.index: Optional[Any] = None
        self.embeddings_matrix: Optional[np.ndarray] = None   # host copy (pairwise path), optional
        self._lookup: Optional[_Lookup] = None
        self._n_dead = 0
        # concurrent single-query searches of one index generation share corpus passes (coalesce.py)
        self.coalesce = True
        self._coalescer: Optional[SearchCoalescer] = None

    @property
    def emb_id_lookup(self) -> Optional[np.ndarray]:
        lk = self._lookup
        return lk.arr if lk is not None else None

    # -- reference surface ------------------------------------------------
    def invalidate(self) -> None:
        """src/svs/kb.py:861-864.  In-flight searches keep the HBM copy alive
        (the handle is reference counted)."""
        _LOG.info("invalidating cached vectors; they'll be re-built next time you `retrieve()`")
        with self._mu:
            idx, self.index = self.index, None
            self.embeddings_matrix = None
            self._lookup = None
            self._n_dead = 0
            self._coalescer = None
        if idx is not None:
            idx.release()

    def _empty_factory(self):
        """(d, reserve) -> an empty index of the configured type, or None if the type cannot grow
        from nothing (test doubles, MultiDeviceIndex)."""
        f, kw = self._index_factory, {}
        if hasattr(f, "func") and hasattr(f, "keywords"):     # functools.partial(DeviceIndex, dtype=...)
            f, kw = f.func, dict(f.keywords)
        make = getattr(f, "empty", None)
        if make is None:
            return None
        return lambda d, reserve: make(d, device=self.device, reserve=reserve, **kw)

    def _build_and_install(self, db):
        """Cold start (src/svs/kb.py:573-618 + :875-876).  Streaming when possible: BLOB -> 32 MiB
        block -> svs_index_append (pinned staging, DMA overlapped with the next block's copy) -- the
        reference's (n, m) host matrix is never materialised; otherwise matrix-then-upload."""
        make = self._empty_factory() if (self._block_builder is not None and not self._keep_host) else None
        if make is None:
            matrix, lookup = self._builder(db)
            return self._install(matrix, lookup)
        holder: dict = {}
        # blocks are decoded straight into the library's pinned staging memory when the index offers it
        acquire = (lambda: holder["idx"].staging_acquire()) if hasattr(getattr(self._index_factory, "func", self._index_factory),
                                                                        "staging_acquire") else None
        try:
            blocks = self._block_builder(db, acquire=acquire)
        except TypeError:            # a block builder without the `acquire` hook
            acquire = None
            blocks = self._block_builder(db)
        n, m = next(blocks)
        if n * m == 0:
            for _ in blocks:
                pass
            return self._install(np.zeros((n, m), dtype=np.float32), np.zeros(n, dtype=np.int64))
        idx = holder["idx"] = make(m, n)
        ids_all = np.empty(n, dtype=np.int64)
        fill = 0
        try:
            for ids, rows in blocks:
                if acquire is not None:
                    idx.staging_commit(len(ids))      # DMA enqueued; the next block is decoded meanwhile
                else:
                    idx.append(rows)                  # copies out of the reused block before returning
                ids_all[fill:fill + len(ids)] = ids
                fill += len(ids)
            if acquire is not None:
                idx.staging_finish()
            assert fill == n and idx.shape == (n, m)
        except BaseException:
            idx.release()
            raise
        self._apply_coalesce(idx)
        with self._mu:
            self.index = idx
            self._lookup = _Lookup(ids_all)
            self._n_dead = 0
            self.embeddings_matrix = None
        return idx

    def _apply_coalesce(self, idx) -> None:
        """Switch the library's coalescing on for a freshly built index (so that the zero-diff
        ``attach()`` path, which reaches the index through np.dot / np.argpartition, gets it too)."""
        if hasattr(idx, "set_coalesce"):
            idx.set_coalesce(bool(self.coalesce))
            idx._coalesce_applied = bool(self.coalesce)

    def _install(self, matrix: np.ndarray, lookup: np.ndarray):
        idx = self._index_factory(matrix, device=self.device)
        self._apply_coalesce(idx)
        with self._mu:
            self.index = idx
            self._lookup = _Lookup(lookup)
            self._n_dead = 0
            self.embeddings_matrix = matrix if self._keep_host else None
        return idx

    # -- incremental update ---------------------------------
    def append(self, new_rows: np.ndarray, new_ids) -> bool:
        """Rows a ``bulk_add_docs`` just committed, in embedding-id order (the order
        ``SELECT id, embedding FROM embeddings`` would return them, src/svs/kb.py:603).
        Edits the HBM copy in place; False if nothing is loaded (the next ``get``
        builds from storage anyway)."""
        new_rows = np.ascontiguousarray(new_rows, dtype=np.float32)
        ids = np.asarray(new_ids, dtype=np.int64)
        with self._mu:
            idx, lk = self.index, self._lookup
            if idx is None or lk is None:
                return False
            if len(ids) == 0:
                return True
            if new_rows.ndim != 2 or new_rows.shape[0] != len(ids) or (len(lk.arr) and ids[0] <= lk.arr[-1]) \
                    or np.any(np.diff(ids) <= 0) or new_rows.shape[1] != idx.shape[1]:
                idx = None   # not a pure append (or dimension change): rebuild
            else:
                idx.append(new_rows)
                lk.arr = np.concatenate([lk.arr, ids])
                if self.embeddings_matrix is not None:
                    self.embeddings_matrix = np.vstack([self.embeddings_matrix, new_rows])
        if idx is None:
            self.invalidate()
            return False
        return True

    def remove(self, emb_ids) -> bool:
        """Embeddings a ``bulk_del_docs`` just committed: their rows are tombstoned in
        HBM (never returned again, other rows keep their indices)."""
        ids = np.asarray(list(emb_ids), dtype=np.int64)
        rebuild = False
        with self._mu:
            idx, lk = self.index, self._lookup
            if idx is None or lk is None:
                return False
            if len(ids) == 0:
                return True
            pos = np.searchsorted(lk.arr, ids)
            if np.any(pos >= len(lk.arr)) or np.any(lk.arr[np.minimum(pos, len(lk.arr) - 1)] != ids):
                rebuild = True
            else:
                self._n_dead += len(ids)
                if self._n_dead > self.COMPACT_AT * len(lk.arr) or self.embeddings_matrix is not None:
                    rebuild = True      # compaction (or a host copy would go stale): rebuild from storage
                else:
                    idx.mask_rows(pos + getattr(idx, "row_offset", 0))
        if rebuild:
            self.invalidate()
            return False
        return True

    def _result(self):
        m = DeviceMatrixView(self.index, self.embeddings_matrix) if self._view else self.embeddings_matrix
        return m, self.emb_id_lookup

    def get_sync(self, db) -> Tuple[Any, np.ndarray]:
        """src/svs/kb.py:866-877."""
        if self.index is not None and self._lookup is not None:
            _LOG.info("using cached vectors")
            return self._result()
        _LOG.info("re-building cached vectors...")
        self._build_and_install(db)
        _LOG.info("re-building cached vectors... DONE!")
        return self._result()

    async def get(self, db) -> Tuple[Any, np.ndarray]:
        """src/svs/kb.py:879-893 (the build runs on an executor thread)."""
        if self.index is not None and self._lookup is not None:
            _LOG.info("using cached vectors")
            return self._result()
        _LOG.info("re-building cached vectors...")
        loop = asyncio.get_running_loop()
        await loop.run_in_executor(None, lambda: self._build_and_install(db))
        _LOG.info("re-building cached vectors... DONE!")
        return self._result()

    # -- the superheavy() body ---------------------------------------------
    def search(self, query_vec: np.ndarray, n: int) -> List[Tuple[float, int]]:
        """``superheavy()`` (src/svs/kb.py:1622-1627): [(score, emb_id)]."""
        idx, lookup, co = self.hold_search()
        try:
            # concurrent callers (threads of a server, AsyncKB's executor threads) share corpus passes
            res = co.search(idx, query_vec, n) if co is not None else idx.search(query_vec, n)
            arr = lookup.arr
            return [(score, int(arr[row])) for score, row in res]
        finally:
            idx.release()

    def search_many(self, query_vecs: np.ndarray, n: int) -> List[List[Tuple[float, int]]]:
        """Batched ``superheavy()``: one result list per row of ``query_vecs``;
        up to 16 (f32) / 32 (f16) queries share one pass over the corpus."""
        idx, lookup = self.hold()
        try:
            scores, rows = idx.search_batch(query_vecs, n)
            arr = lookup.arr
            ids = arr[rows]                                   # (one gather for the whole batch)
            sc = scores.astype(np.float64)
            return [list(zip(sc[i].tolist(), ids[i].tolist())) for i in range(len(scores))]
        finally:
            idx.release()

    def top_pairs(self, n: int) -> List[Tuple[float, int, int]]:
        """``superheavy()`` of document_top_pairwise_scores (src/svs/kb.py:1650-1655):
        [(score, emb_id_1, emb_id_2)]."""
        idx, lookup = self.hold()
        try:
            res = idx.top_pairs(n)
            return [(score, int(lookup.arr[i]), int(lookup.arr[j])) for score, i, j in res]
        finally:
            idx.release()

    def hold(self) -> Tuple[Any, _Lookup]:
        """(index, lookup table) with the caller owning a reference to the index, so a
        concurrent ``invalidate()`` cannot free it mid-search.  Map rows through
        ``lookup.arr`` AFTER the search returns; release the index."""
        with self._mu:
            idx, lookup = self.index, self._lookup
            if idx is None or lookup is None:
                raise RuntimeError("embeddings matrix is not loaded (call get_sync/get first)")
            return idx.share(), lookup

    def hold_search(self) -> Tuple[Any, _Lookup, Optional[SearchCoalescer]]:
        """``hold()`` plus the generation's coalescer (None when switched off): what ``retrieve``
        uses for its single-query search."""
        with self._mu:
            idx, lookup = self.index, self._lookup
            if idx is None or lookup is None:
                raise RuntimeError("embeddings matrix is not loaded (call get_sync/get first)")
            if hasattr(idx, "set_coalesce"):
                # a DeviceIndex coalesces inside the library (svs_index_set_coalesce): the waiting threads
                # block in C, without the GIL -- 64 callers: 20 k queries/s where the Python class below gets 12.6 k
                if getattr(idx, "_coalesce_applied", None) != bool(self.coalesce):
                    idx.set_coalesce(bool(self.coalesce))
                    idx._coalesce_applied = bool(self.coalesce)
                return idx.share(), lookup, None
            if self.coalesce and self._coalescer is None:
                self._coalescer = SearchCoalescer()
            return idx.share(), lookup, (self._coalescer if self.coalesce else None)


# --------------------------------------------------------------------------
# zero-diff attachment through NumPy's __array_function__ protocol
# --------------------------------------------------------------------------
class DeviceScores:
    """What ``np.dot(DeviceMatrixView, q)`` returns: a lazy length-N score
    vector.  ``np.argpartition(self, -k)`` runs the device search."""

    ndim = 1

    def __init__(self, index, query_vec: np.ndarray):
        self._index = index
        self._q = np.asarray(query_vec, dtype=np.float32)
        self._found = {}
        self.shape = (len(index),)
        self.dtype = np.dtype(np.float32)

    def __len__(self) -> int:
        return self.shape[0]

    def __getitem__(self, i):
        i = int(i)
        if i not in self._found:  # not produced by the last search: fetch the full vector once
            full = self._index.scores(self._q)
            self._found = {j: float(v) for j, v in enumerate(full)}
        return np.float32(self._found[i])

    def __array__(self, dtype=None, copy=None):
        out = self._index.scores(self._q)
        return out if dtype is None else out.astype(dtype)

    def __array_function__(self, func, types, args, kwargs):
        if func is np.argpartition and len(args) >= 2 and args[0] is self and not kwargs:
            kth = int(args[1])
            n = self.shape[0]
            if kth < 0 and -kth <= n:
                k = -kth
                res = self._index.search(self._q, k)
                self._found = {row: score for score, row in res}
                rows = np.fromiter((row for _, row in res), dtype=np.int64, count=len(res))
                # argpartition contract: the k largest occupy the last k slots
                out = np.empty(n, dtype=np.int64) if n == k else None
                if out is not None:
                    out[:] = rows[::-1]
                    return out
                return _TailOnly(rows[::-1], n)
        return func(*[np.asarray(a) if isinstance(a, DeviceScores) else a for a in args], **kwargs)


class _TailOnly:
    """Index vector of which only the last k entries are materialised (the only
    ones ``get_top_k`` reads: ``indices[-top_k:]``, src/svs/util.py:202)."""

    def __init__(self, tail: np.ndarray, n: int):
        self._tail, self._n = tail, n

    def __getitem__(self, sl):
        if isinstance(sl, slice) and sl.stop is None and sl.step is None and sl.start is not None \
                and sl.start < 0 and -sl.start <= len(self._tail):
            return self._tail[sl.start:]
        raise IndexError("only the top-k tail of a device argpartition is available")

    def __len__(self) -> int:
        return self._n


class DeviceMatrixView:
    """Stands in for the numpy ``embeddings_matrix`` inside the reference's
    ``retrieve()``.  ``.shape`` serves the log line (kb.py:1629); ``np.dot`` with
    a 1-D query goes to the device; anything else falls back to the host copy
    (e.g. ``np.dot(M, M.T)`` of document_top_pairwise_scores, kb.py:1651)."""

    def __init__(self, index, host_matrix: Optional[np.ndarray]):
        self._index = index
        self._host = host_matrix
        self.shape = tuple(index.shape)
        self.ndim = 2
        self.dtype = np.dtype(np.float32)

    def __len__(self) -> int:
        return self.shape[0]

    @property
    def T(self):
        return self._require_host().T

    def _require_host(self) -> np.ndarray:
        if self._host is None:
            raise RuntimeError("host copy of the embeddings matrix was not kept (keep_host_matrix=False)")
        return self._host

    def __array__(self, dtype=None, copy=None):
        h = self._require_host()
        return h if dtype is None else h.astype(dtype)

    def __array_function__(self, func, types, args, kwargs):
        if func is np.dot and len(args) == 2 and args[0] is self and not kwargs:
            q = np.asarray(args[1])
            if q.ndim == 1:
                if q.shape[0] != self.shape[1] or self.shape[0] == 0:
                    raise ValueError(f"shapes {self.shape} and {q.shape} not aligned: "
                                     f"{self.shape[1]} (dim 1) != {q.shape[0]} (dim 0)")
                return DeviceScores(self._index, q)
        return func(*[np.asarray(a) if isinstance(a, DeviceMatrixView) else a for a in args], **kwargs)


def attach(kb, device: int = 0, keep_host_matrix: bool = True, index_factory: Callable[..., Any] = DeviceIndex,
           devices: Optional[List[int]] = None, dtype: str = "f32"):
    """Swap a reference ``svs.KB`` / ``svs.AsyncKB`` instance's matrix cache for
    the device-backed one.  No reference source line changes; ``retrieve()``
    keeps its exact surface.  ``devices=[0, 1, ...]`` row-shards the corpus over
    several GPUs of this process (``svs_amd.multi.MultiDeviceIndex``); ``dtype``
    ("f32" = the reference's arithmetic, "f16", "fp8") is how the corpus is stored in HBM."""
    import functools
    if devices is not None and len(devices) > 1:
        from .multi import MultiDeviceIndex
        index_factory = functools.partial(MultiDeviceIndex, devices=list(devices), dtype=dtype)
    elif dtype != "f32":
        index_factory = functools.partial(index_factory, dtype=dtype)
    old = kb.embeddings_matrix
    new = DeviceEmbeddingsMatrix(device=device, keep_host_matrix=keep_host_matrix,
                                 index_factory=index_factory, view=True)
    kb.embeddings_matrix = new
    try:
        old.invalidate()
    except Exception:  # noqa: BLE001
        pass
    return kb
