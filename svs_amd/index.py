"""
DeviceIndex: the HBM-resident embedding matrix plus the fused
``np.dot(M, q)`` + ``get_top_k`` search of the reference's ``superheavy()``
closure (reference src/svs/kb.py:1622-1627 / :1184-1189, src/svs/util.py:190-203).

Thin host mirror over the C ABI (include/svs_amd.h).  All arithmetic happens in
the HIP kernels; this module only marshals numpy buffers.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import List, Optional, Tuple

import numpy as np

from . import _native


# HBM element type of the corpus.  "f32" is the reference layout; "f16" rounds
# rows (and queries) to half, f32 accumulate -- BASELINE.json configs[2]/[3];
# "fp8" stores OCP e4m3 with one f32 scale per row -- configs[4].
_DTYPES = {"f32": _native.DTYPE_F32, "f16": _native.DTYPE_F16, "fp8": _native.DTYPE_FP8}


class DeviceIndex:
    """One shard of a corpus, resident in the HBM of one MI355X.

    ``matrix`` is the array ``_Querier.build_embeddings_matrix`` returns
    (reference src/svs/kb.py:600): float32, shape (N, D), C-contiguous.  It is
    copied once (pinned staging -> HBM); the caller keeps ownership.
    """

    def __init__(self, matrix: Optional[np.ndarray], device: int = 0, row_offset: int = 0,
                 dtype: str = "f32", *, _handle: Optional[int] = None):
        self._lib = _native.load()
        self._quick = _native.quick()   # info / retain / release of a per-call reference: under the GIL (see _native.QUICK)
        self._lock = threading.Lock()
        self._h: Optional[int] = None
        if _handle is not None:
            self._h = _handle
        else:
            assert matrix is not None
            m = np.asarray(matrix)
            if m.ndim != 2:
                raise ValueError(f"embeddings matrix must be 2-D, got shape {m.shape}")
            if m.dtype != np.float32 or not m.flags["C_CONTIGUOUS"]:
                m = np.ascontiguousarray(m, dtype=np.float32)
            out = C.c_void_p()
            _native.check(self._lib.svs_index_create(
                m.ctypes.data_as(C.c_void_p), m.shape[0], m.shape[1], _DTYPES[dtype],
                int(device), int(row_offset), C.byref(out)))
            self._h = out.value
        self._refresh()

    def _refresh(self) -> None:
        info = _native.IndexInfo()
        _native.check(self._lib.svs_index_info(self._handle(), C.byref(info)))
        self.n, self.d, self.ld = int(info.n), int(info.d), int(info.ld)
        self.device, self.row_offset = int(info.device), int(info.row_offset)
        self.hbm_bytes = int(info.hbm_bytes)
        self.n_masked = int(info.n_masked)
        self.dtype = {v: k for k, v in _DTYPES.items()}[int(info.dtype)]

    @classmethod
    def from_device_pointer(cls, ptr: int, n: int, d: int, src_ld: Optional[int] = None,
                            device: int = 0, row_offset: int = 0, dtype: str = "f32") -> "DeviceIndex":
        """Corpus rows already in device memory (f32): copies them into the
        index's own HBM layout.  ``ptr`` is a raw device address (e.g.
        ``tensor.data_ptr()``)."""
        lib = _native.load()
        out = C.c_void_p()
        _native.check(lib.svs_index_create_from_device(
            C.c_void_p(ptr), int(n), int(d), int(src_ld if src_ld is not None else d),
            _DTYPES[dtype], int(device), int(row_offset), C.byref(out)))
        return cls(None, _handle=out.value)

    # -- lifetime ---------------------------------------------------------
    @property
    def shape(self) -> Tuple[int, int]:
        return (self.n, self.d)

    def __len__(self) -> int:
        return self.n

    def _handle(self) -> int:
        h = self._h
        if h is None:
            raise RuntimeError("DeviceIndex has been released")
        return h

    def release(self) -> None:
        """Drops this object's reference; in-flight searches on other threads
        keep the HBM alive until they finish (ref-counted in the library)."""
        with self._lock:
            h, self._h = self._h, None
        if h is not None:
            self._lib.svs_index_release(h)

    close = release

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def _live_rows(self) -> int:
        """Rows a search can return right now (the handle may have been appended to or
        masked through another owner since this object last looked)."""
        info = _native.IndexInfo()
        _native.check(self._quick.svs_index_info(self._handle(), C.byref(info)))
        return max(int(info.n) - int(info.n_masked), 0)

    def _pinned_handle(self) -> int:
        """retain() under the lock so a concurrent release() cannot free the
        handle between reading it and entering the library."""
        with self._lock:
            h = self._handle()
            self._quick.svs_index_retain(h)
            return h

    def _unpin(self, h: int) -> None:
        """Drops the reference _pinned_handle() took (normally not the last one: the owner holds its own)."""
        self._quick.svs_index_release(h)

    # -- incremental update ---------------------------------
    def append(self, matrix: np.ndarray) -> None:
        """New rows behind the existing ones (what a rebuilt matrix would hold after
        ``bulk_add_docs``), without re-uploading the corpus."""
        m = np.ascontiguousarray(matrix, dtype=np.float32)
        if m.ndim != 2 or (m.shape[0] and m.shape[1] != self.d):
            raise ValueError(f"cannot append shape {m.shape} to an index of dimension {self.d}")
        _native.check(self._lib.svs_index_append(self._handle(), m.ctypes.data_as(C.c_void_p), m.shape[0]))
        self._refresh()

    @classmethod
    def empty(cls, d: int, device: int = 0, row_offset: int = 0, dtype: str = "f32",
              reserve: int = 0) -> "DeviceIndex":
        """An index of dimension ``d`` with no rows yet (optionally with room for ``reserve`` rows),
        to be filled with ``append`` / ``append_device`` block by block -- the way a cold start
        streams BLOBs out of SQLite without ever holding the whole matrix on the host."""
        lib = _native.load()
        out = C.c_void_p()
        _native.check(lib.svs_index_create(None, 0, int(d), _DTYPES[dtype], int(device), int(row_offset), C.byref(out)))
        idx = cls(None, _handle=out.value)
        if reserve > 0:
            idx.reserve(reserve)
        return idx

    def reserve(self, rows: int) -> None:
        _native.check(self._lib.svs_index_reserve(self._handle(), int(rows)))
        self._refresh()

    def append_device(self, ptr: int, n: int, src_ld: Optional[int] = None) -> None:
        """``append`` for f32 rows already in this device's memory (raw device address)."""
        _native.check(self._lib.svs_index_append_from_device(
            self._handle(), C.c_void_p(ptr), int(n), int(src_ld if src_ld is not None else self.d)))
        self._refresh()

    # -- cold start: BLOBs decoded straight into the library's pinned staging blocks --------------
    def staging_acquire(self) -> np.ndarray:
        """A (rows_cap, d) float32 array over pinned memory owned by the library: fill its first rows,
        then ``staging_commit(n)``.  Two blocks alternate; the DMA of one overlaps the filling of the other."""
        ptr, cap = C.c_void_p(), C.c_int64(0)
        _native.check(self._lib.svs_index_staging_acquire(self._handle(), C.byref(ptr), C.byref(cap)))
        buf = (C.c_float * (cap.value * self.d)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=np.float32).reshape(cap.value, self.d)

    def staging_commit(self, n_rows: int) -> None:
        _native.check(self._lib.svs_index_staging_commit(self._handle(), int(n_rows)))
        self._refresh()      # the library publishes the new row count at commit: keep the wrapper's in step

    def staging_finish(self) -> None:
        _native.check(self._lib.svs_index_staging_finish(self._handle()))
        self._refresh()

    def mask_rows(self, rows) -> None:
        """Tombstone rows (global indices): they are never returned again; the other
        rows keep their indices."""
        r = np.ascontiguousarray(rows, dtype=np.int64)
        _native.check(self._lib.svs_index_mask_rows(self._handle(), r.ctypes.data_as(C.c_void_p), r.shape[0]))
        self._refresh()

    def share(self) -> "DeviceIndex":
        """A second owner of the same HBM copy (its own reference): what an
        in-flight AsyncKB search holds so that ``invalidate()`` on another task
        cannot pull the corpus from under it (reference semantics: the closure
        keeps the numpy arrays alive, src/svs/kb.py:1180-1190)."""
        return DeviceIndex(None, _handle=self._pinned_handle())

    # -- search -----------------------------------------------------------
    def search_batch(self, queries: np.ndarray, n: int) -> Tuple[np.ndarray, np.ndarray]:
        """Top-n for each row of ``queries`` (nq, D).  Returns
        (scores f32 (nq, count), rows i64 (nq, count)), count = min(max(n,0), N),
        each row ordered (score desc, row desc)."""
        assert isinstance(n, int)  # reference src/svs/util.py:197
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim != 2:
            raise ValueError(f"queries must be 2-D, got shape {q.shape}")
        nq, d = q.shape
        h = self._pinned_handle()
        try:
            # clamp before allocating and before the int32 argument (reference src/svs/util.py:198-199
            # clamps top_k to len(scores) first): k = 2**32 must mean "rank everything", not 0.  (The row count is
            # read through the pinned handle: the index may have been appended to or masked through another owner.)
            info = _native.IndexInfo()
            _native.check(self._quick.svs_index_info(h, C.byref(info)))
            k = min(max(n, 0), max(int(info.n) - int(info.n_masked), 0))
            scores = np.empty((nq, k), dtype=np.float32)
            rows = np.empty((nq, k), dtype=np.int64)
            count = C.c_int32(0)
            # (the one call of a search that releases the GIL)
            _native.check(self._lib.svs_index_search(h, q.ctypes.data, nq, d, k, scores.ctypes.data, rows.ctypes.data,
                                                     C.byref(count)))
        finally:
            self._unpin(h)
        c = count.value
        return (scores, rows) if c == k else (scores[:, :c], rows[:, :c])

    def search(self, query_vec: np.ndarray, n: int) -> List[Tuple[float, int]]:
        """``get_top_k(np.dot(M, query_vec), n)``: list of (score, row index),
        python float / python int, as the reference returns them."""
        assert isinstance(n, int)
        q = np.asarray(query_vec, dtype=np.float32)
        if q.ndim != 1:
            raise ValueError(f"query must be 1-D, got shape {q.shape}")
        s, r = self.search_batch(q[None, :], n)
        # (tolist(): python floats -- the exact widening float(np.float32) does, src/svs/util.py:203 -- and ints in C)
        return list(zip(s[0].astype(np.float64).tolist(), r[0].tolist()))

    def scores(self, query_vec: np.ndarray) -> np.ndarray:
        """The raw ``np.dot(M, q)`` vector, f32 (N,)."""
        q = np.ascontiguousarray(query_vec, dtype=np.float32)
        if q.ndim != 1:
            raise ValueError(f"query must be 1-D, got shape {q.shape}")
        h = self._pinned_handle()
        try:
            # The library writes one f32 per row the HANDLE holds when the call runs, never more than the capacity
            # passed: if rows were appended (another thread, another owner of the handle) since the buffer was
            # sized, the call fails, reports the new count, and is repeated with a buffer of that size.
            rows = self.n
            for _ in range(8):
                out = np.empty(rows, dtype=np.float32)
                now = C.c_int64(0)
                rc = self._lib.svs_index_scores_n(h, q.ctypes.data_as(C.c_void_p), q.shape[0],
                                                  out.ctypes.data_as(C.c_void_p), rows, C.byref(now))
                if rc == _native.SVS_OK:
                    return out[:now.value]
                if now.value <= rows:
                    break
                rows = now.value
            _native.check(rc)
            raise RuntimeError("svs_index_scores_n: the index kept growing")
        finally:
            self._unpin(h)

    def search_device(self, q_ptr: int, nq: int, d: int, k: int, out_scores_ptr: int,
                      out_rows_ptr: int, stream: int = 0) -> int:
        """Device-pointer variant (no synchronisation): see svs_index_search_device."""
        count = C.c_int32(0)
        h = self._pinned_handle()
        try:
            _native.check(self._lib.svs_index_search_device(
                h, C.c_void_p(q_ptr), int(nq), int(d), int(k), C.c_void_p(out_scores_ptr),
                C.c_void_p(out_rows_ptr), C.byref(count), C.c_void_p(stream)))
        finally:
            self._unpin(h)
        return count.value

    def top_pairs(self, n: int) -> List[Tuple[float, int, int]]:
        """``get_top_pairs(np.dot(M, M.T), n)`` (reference src/svs/kb.py:1651,
        src/svs/util.py:206-233): [(score, row_i, row_j)] with i < j."""
        assert isinstance(n, int)
        live = self._live_rows()
        k = min(max(n, 0), live * (live - 1) // 2)   # reference src/svs/util.py:198-199 on the pair list
        scores = np.empty(k, dtype=np.float32)
        ri = np.empty(k, dtype=np.int64)
        rj = np.empty(k, dtype=np.int64)
        count = C.c_int32(0)
        h = self._pinned_handle()
        try:
            _native.check(self._lib.svs_index_top_pairs(
                h, k, scores.ctypes.data_as(C.c_void_p), ri.ctypes.data_as(C.c_void_p),
                rj.ctypes.data_as(C.c_void_p), C.byref(count)))
        finally:
            self._unpin(h)
        c = count.value
        return [(float(s), int(a), int(b)) for s, a, b in zip(scores[:c], ri[:c], rj[:c])]

    # -- parity support -----------------------------------------------------
    def stored_rows(self, row0: int = 0, nrows: Optional[int] = None) -> np.ndarray:
        """Rows exactly as held in HBM, dequantised to f32 (the corpus the oracle
        must be run on for the f16 / fp8 dtypes)."""
        nrows = self.n - row0 if nrows is None else nrows
        out = np.empty((nrows, self.d), dtype=np.float32)
        _native.check(self._lib.svs_index_debug_dequant(self._handle(), int(row0), int(nrows),
                                                        out.ctypes.data_as(C.c_void_p)))
        return out

    def stored_query(self, query_vec: np.ndarray) -> np.ndarray:
        """The query as this index's kernels see it (rounded / quantised)."""
        q = np.ascontiguousarray(query_vec, dtype=np.float32)
        out = np.empty_like(q)
        _native.check(self._lib.svs_index_debug_query(self._handle(), q.ctypes.data_as(C.c_void_p), q.shape[0],
                                                      out.ctypes.data_as(C.c_void_p)))
        return out

    # -- measurement ------------------------------------------------------
    def set_timing(self, enable) -> None:
        """True / 1: time every search; N: every N-th; False / 0: off."""
        _native.check(self._lib.svs_index_set_timing(self._handle(), int(enable)))

    def get_timing(self) -> Tuple[float, float, int]:
        t = _native.Timing()
        _native.check(self._lib.svs_index_get_timing(self._handle(), C.byref(t)))
        self.last_dominant_ms_sum = float(t.dominant_ms_sum)   # the dominant kernel alone (bench.py's roofline)
        return float(t.score_ms_sum), float(t.select_ms_sum), int(t.launches)

    def set_coalesce(self, enable: bool) -> None:
        """Concurrent single-query searches on this handle share corpus passes (svs_index_set_coalesce)."""
        _native.check(self._lib.svs_index_set_coalesce(self._handle(), 1 if enable else 0))

    def coalesce_stats(self) -> Tuple[int, int]:
        """(corpus passes, queries answered) through the coalescing path."""
        p, q = C.c_int64(0), C.c_int64(0)
        _native.check(self._lib.svs_index_coalesce_stats(self._handle(), C.byref(p), C.byref(q)))
        return p.value, q.value

    def coalesce_hold(self, n: int) -> None:
        """The next coalesced pass waits (at most 5 s) for n queued callers (svs_internal_coalesce_hold; tests)."""
        _native.check(self._lib.svs_internal_coalesce_hold(self._handle(), int(n)))

    def coalesce_sizes(self) -> dict:
        """{queries per pass: passes} of the coalescing path (svs_index_coalesce_sizes)."""
        out = (C.c_int64 * 257)()
        _native.check(self._lib.svs_index_coalesce_sizes(self._handle(), out, 257))
        return {s: int(out[s]) for s in range(257) if out[s]}

    def set_variant(self, variant: int) -> None:
        _native.check(self._lib.svs_index_set_variant(self._handle(), int(variant)))
