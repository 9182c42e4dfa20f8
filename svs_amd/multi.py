"""
MultiDeviceIndex: one process, several MI355X, the corpus row-sharded across them
(the in-process form of the C-ABI sketch's ``devices, ndev``).

For a single Python process that owns all GPUs of a node -- the way ``svs.KB`` is
normally used -- this needs no launcher and no RCCL: shard g holds the contiguous
row block ``shard_bounds(N, G, g)`` on device ``devices[g]`` with
``row_offset = lo``; a search runs on all shards concurrently (ctypes releases the
GIL inside the C ABI, so one thread per shard is real parallelism), every shard
returns its local top-k with GLOBAL rows (1.2 KB each, written zero-copy into pinned
host memory), and the host merges them under the same total order
(score desc, row desc).  The merged result is identical to a single-device index:
a row's score does not depend on where the row lives.

(The one-process-per-GPU form over torch.distributed/RCCL is ``svs_amd.sharded``.)
"""
from __future__ import annotations

import threading
from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .index import DeviceIndex
from .sharded import merge_topk_batch, shard_bounds


class _SharedPool:
    """One worker per shard, shared by an index and its share()d copies (share() runs on every
    AsyncKB search: it must not spawn threads); shut down when the last holder lets go."""

    def __init__(self, workers: int):
        self.ex = ThreadPoolExecutor(max_workers=max(1, workers))
        self._refs = 1
        self._mu = threading.Lock()

    def retain(self) -> "_SharedPool":
        with self._mu:
            self._refs += 1
        return self

    def release(self) -> None:
        with self._mu:
            self._refs -= 1
            last = self._refs == 0
        if last:
            self.ex.shutdown(wait=False)


class MultiDeviceIndex:
    """Has the surface of ``DeviceIndex`` that the KB layer uses (search,
    search_batch, scores, append, mask_rows, share, release, shape)."""

    def __init__(self, matrix: Optional[np.ndarray], devices: Sequence[int] = (0,), device: Optional[int] = None,
                 dtype: str = "f32", *, _shards: Optional[List[DeviceIndex]] = None, _bounds=None, _pool=None):
        if _shards is not None:
            self._shards, self._bounds = _shards, list(_bounds)
        else:
            m = np.ascontiguousarray(matrix, dtype=np.float32)
            if m.ndim != 2:
                raise ValueError(f"embeddings matrix must be 2-D, got shape {m.shape}")
            devices = list(devices)
            g = len(devices)
            self._bounds = [shard_bounds(m.shape[0], g, r) for r in range(g)]
            self._shards = []
            try:
                # uploads run concurrently: each shard has its own pinned staging and stream
                with ThreadPoolExecutor(max_workers=g) as ex:
                    futs = [ex.submit(DeviceIndex, m[lo:hi], devices[r], lo, dtype) for r, (lo, hi) in enumerate(self._bounds)]
                    for f in futs:
                        self._shards.append(f.result())
            except BaseException:
                for s in self._shards:
                    s.release()
                raise
        self._pool_ref = _pool if _pool is not None else _SharedPool(len(self._shards))
        self._pool = self._pool_ref.ex
        self._mu = threading.Lock()
        self.d = self._shards[0].d
        self.dtype = self._shards[0].dtype
        self.row_offset = 0

    # -- geometry -----------------------------------------------------------
    @property
    def n(self) -> int:
        return sum(s.n for s in self._shards)

    @property
    def shape(self) -> Tuple[int, int]:
        return (self.n, self.d)

    def __len__(self) -> int:
        return self.n

    @property
    def n_masked(self) -> int:
        return sum(s.n_masked for s in self._shards)

    @property
    def devices(self) -> List[int]:
        return [s.device for s in self._shards]

    # -- lifetime -------------------------------------------------------------
    def share(self) -> "MultiDeviceIndex":
        return MultiDeviceIndex(None, _shards=[s.share() for s in self._shards], _bounds=self._bounds, _pool=self._pool_ref.retain())

    def release(self) -> None:
        for s in self._shards:
            s.release()
        self._pool_ref.release()

    close = release

    # -- search ---------------------------------------------------------------
    def search_batch(self, queries: np.ndarray, n: int) -> Tuple[np.ndarray, np.ndarray]:
        assert isinstance(n, int)
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim != 2:
            raise ValueError(f"queries must be 2-D, got shape {q.shape}")
        live = [s for s in self._shards if s.n > 0]
        if not live or q.shape[1] != self.d:
            # same error as numpy / a single index
            return self._shards[0].search_batch(q, n)
        parts = list(self._pool.map(lambda s: s.search_batch(q, n), live))
        k = max(n, 0)
        count = min(k, self.n - self.n_masked)
        # shards return min(k, their live rows) each: pad to a common width, then ONE lexsort over
        # the (nq, G * width) table (was: a Python loop of nq sorts)
        width = max(p[0].shape[1] for p in parts)
        sc = np.full((len(parts), q.shape[0], width), -np.inf, dtype=np.float32)
        rw = np.full((len(parts), q.shape[0], width), -1, dtype=np.int64)
        for g, (ps, pr) in enumerate(parts):
            sc[g, :, : ps.shape[1]] = ps
            rw[g, :, : pr.shape[1]] = pr
        return merge_topk_batch(sc, rw, count)

    def search(self, query_vec: np.ndarray, n: int) -> List[Tuple[float, int]]:
        assert isinstance(n, int)
        q = np.asarray(query_vec, dtype=np.float32)
        if q.ndim != 1:
            raise ValueError(f"query must be 1-D, got shape {q.shape}")
        s, r = self.search_batch(q[None, :], n)
        return [(float(a), int(b)) for a, b in zip(s[0], r[0])]

    def scores(self, query_vec: np.ndarray) -> np.ndarray:
        parts = list(self._pool.map(lambda s: s.scores(query_vec), [s for s in self._shards if s.n > 0]))
        return np.concatenate(parts) if parts else np.empty(0, dtype=np.float32)

    # -- incremental update -----------------------------------------------------
    def append(self, matrix: np.ndarray) -> None:
        """New rows go behind the LAST shard (global row = previous N + i), so existing
        global row numbers -- and the caller's emb_id_lookup -- stay valid."""
        with self._mu:
            self._shards[-1].append(matrix)
            lo, _ = self._bounds[-1]
            self._bounds[-1] = (lo, lo + self._shards[-1].n)

    def mask_rows(self, rows) -> None:
        r = np.asarray(rows, dtype=np.int64)
        with self._mu:
            for s, (lo, hi) in zip(self._shards, self._bounds):
                mine = r[(r >= lo) & (r < hi)]
                if len(mine):
                    s.mask_rows(mine)
            if np.any((r < 0) | (r >= self.n)):
                raise ValueError("row out of range")

    def top_pairs(self, n: int):
        raise NotImplementedError("pairwise scores need the whole corpus on one device; build a DeviceIndex for it")

    def stored_rows(self, row0: int = 0, nrows: Optional[int] = None) -> np.ndarray:
        full = np.concatenate([s.stored_rows() for s in self._shards if s.n > 0])
        return full[row0:(None if nrows is None else row0 + nrows)]

    def stored_query(self, query_vec: np.ndarray) -> np.ndarray:
        return self._shards[0].stored_query(query_vec)


class NativeMultiIndex:
    """The same thing behind the C ABI (``svs_multi_*``, include/svs_amd.h): sharding, the per-shard
    worker threads and the merge live in the library, so a caller that is not Python gets them too.
    Search surface of ``DeviceIndex`` (search, search_batch, shape, release)."""

    def __init__(self, matrix: np.ndarray, devices: Sequence[int] = (0,), dtype: str = "f32"):
        import ctypes as C
        from . import _native
        from .index import _DTYPES
        m = np.ascontiguousarray(matrix, dtype=np.float32)
        if m.ndim != 2:
            raise ValueError(f"embeddings matrix must be 2-D, got shape {m.shape}")
        self._lib = _native.load()
        self._C, self._native = C, _native
        dev = (C.c_int32 * len(devices))(*[int(x) for x in devices])
        out = C.c_void_p()
        _native.check(self._lib.svs_multi_create(m.ctypes.data_as(C.c_void_p) if m.size else None, m.shape[0], m.shape[1],
                                                 _DTYPES[dtype], dev, len(devices), C.byref(out)))
        self._h = out.value
        self.d, self.dtype = int(m.shape[1]), dtype

    def _info(self):
        C = self._C
        g, n, d, dead = C.c_int32(), C.c_int64(), C.c_int32(), C.c_int64()
        # (through the GIL-holding binding: a few loads, no lock -- _native.QUICK)
        self._native.check(self._native.quick().svs_multi_info(self._h, C.byref(g), C.byref(n), C.byref(d), C.byref(dead)))
        return g.value, n.value, d.value, dead.value

    @property
    def n(self) -> int:
        return self._info()[1]

    @property
    def shape(self) -> Tuple[int, int]:
        return (self.n, self.d)

    def search_batch(self, queries: np.ndarray, n: int) -> Tuple[np.ndarray, np.ndarray]:
        assert isinstance(n, int)
        C = self._C
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim != 2:
            raise ValueError(f"queries must be 2-D, got shape {q.shape}")
        _, rows, _, dead = self._info()
        k = min(max(n, 0), max(rows - dead, 0), 2 ** 31 - 1)
        scores = np.empty((q.shape[0], k), dtype=np.float32)
        out_rows = np.empty((q.shape[0], k), dtype=np.int64)
        count = C.c_int32(0)
        self._native.check(self._lib.svs_multi_search(self._h, q.ctypes.data_as(C.c_void_p), q.shape[0], q.shape[1], k,
                                                      scores.ctypes.data_as(C.c_void_p), out_rows.ctypes.data_as(C.c_void_p), C.byref(count)))
        return scores[:, :count.value], out_rows[:, :count.value]

    def search(self, query_vec: np.ndarray, n: int) -> List[Tuple[float, int]]:
        assert isinstance(n, int)
        q = np.asarray(query_vec, dtype=np.float32)
        if q.ndim != 1:
            raise ValueError(f"query must be 1-D, got shape {q.shape}")
        s, r = self.search_batch(q[None, :], n)
        return [(float(a), int(b)) for a, b in zip(s[0], r[0])]

    def mask_rows(self, rows) -> None:
        """Tombstones GLOBAL rows: each goes to the shard that holds it."""
        C = self._C
        g, n, _, _ = self._info()
        per = (n + g - 1) // g if g else 0
        by_shard = {}
        for r in rows:
            by_shard.setdefault(min(int(r) // per, g - 1) if per else 0, []).append(int(r))
        for sg, rr in by_shard.items():
            h = C.c_void_p()
            self._native.check(self._lib.svs_multi_shard(self._h, sg, C.byref(h)))
            try:
                glob = np.asarray(rr, dtype=np.int64)   # (svs_index_mask_rows takes global rows: the shard knows its offset)
                self._native.check(self._lib.svs_index_mask_rows(h, glob.ctypes.data_as(C.c_void_p), len(glob)))
            finally:
                self._lib.svs_index_release(h)

    def set_coalesce(self, enable: bool) -> None:
        """Concurrent single-query searches share passes over all shards (svs_multi_set_coalesce)."""
        self._native.check(self._lib.svs_multi_set_coalesce(self._h, 1 if enable else 0))

    def coalesce_stats(self) -> Tuple[int, int]:
        C = self._C
        p, q = C.c_int64(0), C.c_int64(0)
        self._native.check(self._lib.svs_multi_coalesce_stats(self._h, C.byref(p), C.byref(q)))
        return p.value, q.value

    def release(self) -> None:
        if self._h:
            self._lib.svs_multi_release(self._h)
            self._h = None

    close = release

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass
