"""
Row-sharded search across GPUs.

One process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI on
ROCm, "gloo" for CPU tests).  Rank r holds the contiguous row block
``[r*ceil(N/G), min((r+1)*ceil(N/G), N))`` of the C-contiguous matrix
(reference layout, src/svs/kb.py:600), so ``global row = local row + offset`` and
``emb_id_lookup`` (src/svs/kb.py:601,1626) stays host-side, unsharded.

Per query every rank produces its LOCAL top-k (k, not k/G: all winners may sit
in one shard) with global row indices; the only exchange step is one gather of
``G * k * (f32 score + i64 row)`` (9.6 KB at G=8, k=100 -- latency-bound, xGMI
bandwidth is irrelevant), followed by a merge under the same total order
(score desc, row desc) as the single-GPU path.  Because every row's score is
computed by the same kernel with the same summation order wherever the row
lives, the merged result is identical for G in {1, 2, 4, 8}.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row block of ``rank``."""
    per = -(-n // world) if world > 0 else n
    lo = min(rank * per, n)
    hi = min(lo + per, n)
    return lo, hi


def merge_topk(scores: np.ndarray, rows: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """Host merge (H1): ``scores``/``rows`` are (G*k',) candidate lists from all
    shards (padding entries have row < 0).  Returns the best ``k`` under
    (score desc, row desc) -- the order of reference src/svs/util.py:203."""
    scores = np.asarray(scores, dtype=np.float32).ravel()
    rows = np.asarray(rows, dtype=np.int64).ravel()
    live = rows >= 0
    scores, rows = scores[live], rows[live]
    # -0.0 and +0.0 are one score (python compares them equal): fold before ordering
    key = scores + np.float32(0.0)
    order = np.lexsort((rows, key))[::-1][: max(k, 0)]
    return scores[order], rows[order]


def merge_topk_batch(scores: np.ndarray, rows: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """``merge_topk`` for nq queries at once: ``scores``/``rows`` are (G, nq, k') candidate lists
    (padding entries: score -inf, row -1).  Returns (nq, k) arrays; requires every query to have >= k live
    candidates (true whenever k <= rows in the corpus: each shard returns min(k', its rows)).

    Global rows below 2^32 (every corpus the keyed GPU paths take): the candidates are ordered the way the kernels
    order them -- ONE sort of 64-bit keys, orderable score bits above the row (csrc/keys.h) -- 10 us per query for
    8 x 100 candidates where a two-key ``np.lexsort`` takes 56-82 us; beyond, the lexsort."""
    g, nq, kk = scores.shape
    sc = np.ascontiguousarray(np.transpose(scores, (1, 0, 2))).reshape(nq, g * kk).astype(np.float32, copy=False)
    rw = np.ascontiguousarray(np.transpose(rows, (1, 0, 2))).reshape(nq, g * kk).astype(np.int64, copy=False)
    k = max(k, 0)
    if rw.size and int(rw.max()) < (1 << 32):
        b = (sc + np.float32(0.0)).view(np.uint32)                    # -0.0 == +0.0 (python compares them equal)
        key32 = np.where(b >> 31, ~b, b | np.uint32(0x80000000))      # larger float <=> larger key
        nan = np.isnan(sc)
        if nan.any():
            key32[nan] = np.uint32(0xFFFFFFFF)                        # any NaN on top (np.argpartition sorts NaN last == largest), as keys.h
        key = (key32.astype(np.uint64) << np.uint64(32)) | (rw.astype(np.uint64) & np.uint64(0xFFFFFFFF))
        key[rw < 0] = 0                                               # padding loses against every real row
        key.sort(axis=1)
        top = key[:, ::-1][:, :k]
        k32 = (top >> np.uint64(32)).astype(np.uint32)
        bits = np.where(k32 >> 31, k32 & np.uint32(0x7FFFFFFF), ~k32)
        bits[k32 == np.uint32(0xFFFFFFFF)] = np.uint32(0x7FC00000)
        # (the scores handed back are the candidates' own bits: the kernels never emit -0.0, keys.h folds it)
        return np.ascontiguousarray(bits).view(np.float32), (top & np.uint64(0xFFFFFFFF)).astype(np.int64)
    key = sc + np.float32(0.0)
    # padding (row -1) must lose against a real row with the same score (-inf): row is the tie-break anyway
    order = np.lexsort((rw, key), axis=-1)[:, ::-1][:, :k]
    return np.take_along_axis(sc, order, axis=1), np.take_along_axis(rw, order, axis=1)


# ---- wire format of one rank's local top-k ---------------------------------------
# One record = the result of ONE search call of nq queries:
#   [nq*k f32 scores | pad to 8 B | nq*k i64 global rows];  entries past a shard's count carry
# score -inf, row -1.  It is exactly what svs_index_search_device writes (scores at the record's
# start, rows at `record_layout()[0]`), so the GPU path never touches the result on the host
# before the exchange, and ONE collective moves both arrays.
def record_layout(k: int, nq: int = 1) -> Tuple[int, int]:
    """(offset of the row array, record bytes)."""
    s_bytes = (nq * k * 4 + 7) // 8 * 8
    return s_bytes, s_bytes + nq * k * 8


def unpack_records(buf: np.ndarray, world: int, k: int, nq: int = 1) -> Tuple[np.ndarray, np.ndarray]:
    """``buf``: uint8 (world, record_bytes) -> (scores f32 (world, nq*k), rows i64 (world, nq*k))."""
    s_bytes, rec = record_layout(k, nq)
    buf = np.ascontiguousarray(buf, dtype=np.uint8).reshape(world, rec)
    sc = np.ascontiguousarray(buf[:, : nq * k * 4]).view(np.float32).reshape(world, nq * k)
    rw = np.ascontiguousarray(buf[:, s_bytes:]).view(np.int64).reshape(world, nq * k)
    return sc, rw


def pack_record(scores: np.ndarray, rows: np.ndarray, k: int, nq: int = 1) -> np.ndarray:
    """Host-side packer (host/CPU path and tests): ``scores``/``rows`` (nq, c) with c <= k, or 1-D
    for one query.  The GPU path has the search kernel write both arrays straight into the record."""
    s_bytes, rec = record_layout(k, nq)
    out = np.zeros(rec, dtype=np.uint8)
    sc = np.full((nq, k), -np.inf, dtype=np.float32)
    rw = np.full((nq, k), -1, dtype=np.int64)
    scores = np.asarray(scores, dtype=np.float32).reshape(nq, -1)
    rows = np.asarray(rows, dtype=np.int64).reshape(nq, -1)
    sc[:, : scores.shape[1]] = scores
    rw[:, : rows.shape[1]] = rows
    out[: nq * k * 4] = sc.reshape(-1).view(np.uint8)
    out[s_bytes:] = rw.reshape(-1).view(np.uint8)
    return out


class ShardedIndex:
    """One process per GPU: rank r holds rows ``shard_bounds(N, G, r)``; this class owns the
    exchange + merge.  ``local`` is either

    * a ``DeviceIndex`` built with ``row_offset = shard_bounds()[0]`` plus a torch ``device``:
      the search kernel writes the packed record into HBM, ONE ``all_gather_into_tensor`` (RCCL)
      moves the records of every rank, ``dst`` copies them home and merges; or
    * a callable ``local_search(queries (nq,d) f32, k) -> (scores (nq,c), rows (nq,c) GLOBAL)``:
      the host path (CPU tests over gloo; any backend whose results are numpy arrays).

    Two ways to drive it: ``search`` / ``search_batch`` (blocking, one record, one collective per
    call) and the pipelined pair ``open`` + ``enqueue`` / ``collect`` that ``bench.py`` measures
    (single-query searches alternate between HIP streams, ``gather_every`` records share one
    collective, ``dst`` streams them back with an async copy).  Results arrive on ``dst`` (None
    elsewhere) and are identical for every G: a row's score does not depend on where it lives."""

    def __init__(self, local, n_total: int, group=None, dst: int = 0, device=None, *,
                 gather_every: int = 8, streams: int = 2, force_collective: bool = False):
        import torch.distributed as dist
        self._dist = dist
        self.local = local
        self._on_device = device is not None and hasattr(local, "search_device")
        self.local_search = None if self._on_device else (local.search_batch if hasattr(local, "search_batch") else local)
        self.n_total = int(n_total)
        self.group = group
        self.dst = dst
        self.device = device
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # one rank needs no exchange (records go straight to pinned host memory); `force_collective` takes the
        # N > 1 path anyway -- records in HBM, RCCL all-gather, async copy home -- so that the real backend can be
        # exercised on a one-GPU box (tests)
        self._multi = self.world > 1 or (bool(force_collective) and dist.is_initialized())
        self.gather_every = max(1, int(gather_every)) if self._multi else 1
        self._n_streams = max(1, int(streams)) if not self._multi else max(2, int(streams))
        self._pipe = None
        self._rec_cache = {}

    # ---- one record, one collective -------------------------------------------------
    def _exchange(self, rec):
        """rec: uint8 tensor (record bytes) of this rank -> uint8 (world, record bytes) on dst, else None."""
        import torch
        if not self._multi:
            return rec[None]
        out = torch.empty((self.world, rec.numel()), dtype=torch.uint8, device=rec.device)
        self._dist.all_gather_into_tensor(out.view(-1), rec, group=self.group)
        return out if self.rank == self.dst else None

    def search_batch(self, queries: np.ndarray, n: int) -> Optional[Tuple[np.ndarray, np.ndarray]]:
        import torch
        assert isinstance(n, int)
        q = np.ascontiguousarray(queries, dtype=np.float32)
        nq = q.shape[0]
        k = max(n, 0)
        count = min(k, self.n_total)
        if k == 0 or nq == 0:
            return (np.empty((nq, 0), np.float32), np.empty((nq, 0), np.int64)) if self.rank == self.dst else None
        s_bytes, rec_bytes = record_layout(k, nq)
        if self._on_device:
            qt = torch.from_numpy(q).to(self.device)
            rec = torch.empty(rec_bytes, dtype=torch.uint8, device=self.device)
            st = torch.cuda.current_stream(self.device)
            self.local.search_device(qt.data_ptr(), nq, q.shape[1], k, rec.data_ptr(), rec.data_ptr() + s_bytes, st.cuda_stream)
            got = self._exchange(rec)
            if got is None:
                return None
            buf = got.cpu().numpy()                     # (synchronises this stream)
        else:
            ls, lr = self.local_search(q, k)
            rec = torch.from_numpy(pack_record(ls, lr, k, nq))
            if self.device is not None:
                rec = rec.to(self.device)
            got = self._exchange(rec)
            if got is None:
                return None
            buf = got.cpu().numpy()
        sc, rw = unpack_records(buf, self.world, k, nq)
        return merge_topk_batch(sc.reshape(self.world, nq, k), rw.reshape(self.world, nq, k), count)

    def search(self, query_vec: np.ndarray, n: int) -> Optional[List[Tuple[float, int]]]:
        res = self.search_batch(np.asarray(query_vec, dtype=np.float32)[None, :], n)
        if res is None:
            return None
        return [(float(a), int(b)) for a, b in zip(res[0][0], res[1][0])]

    # ---- pipelined single-query searches (device path; what bench.py times) ----------------
    def open(self, capacity: int, k: int) -> None:
        """Buffers for up to ``capacity`` enqueued single-query searches of top-``k``."""
        import torch
        assert self._on_device, "the pipelined path needs a DeviceIndex and a torch device"
        s_bytes, rec = record_layout(k)
        g = self.gather_every
        nchunks = (capacity + g - 1) // g
        dev = self.device
        p = {"k": k, "rec": rec, "s_bytes": s_bytes, "n": 0, "sent": 0, "nchunks": nchunks,
             # N = 1: the final top-k kernel writes the record straight into pinned host memory
             # (zero-copy, no D2H).  N > 1: records stay in HBM for the RCCL all-gather; dst streams
             # each gathered chunk home with an async copy.
             "host": torch.zeros((nchunks, self.world, g * rec), dtype=torch.uint8, pin_memory=True),
             "local": None if not self._multi else torch.zeros((nchunks, g * rec), device=dev, dtype=torch.uint8),
             "gathered": None if not self._multi else torch.zeros((nchunks, self.world, g * rec), device=dev, dtype=torch.uint8),
             "streams": [torch.cuda.Stream(device=dev) for _ in range(self._n_streams)],
             # the exchange has a stream of its own: issued on a search stream (rounds 1-3) that stream's next searches
             # sat behind the collective and the copy home -- ~120 us per exchange with one of two streams stalled,
             # 15 us per step at one GPU's share of the 8-GPU strong-scaling run (125 k rows: 0.131 -> see DESIGN 6)
             "xstream": None if not self._multi else torch.cuda.Stream(device=dev), "done": {}}
        self._pipe = p

    def _send_chunk(self, c: int, st) -> None:
        """One all-gather for chunk c (its records were written on every search stream), issued on the exchange
        stream behind an event on each of them; the search streams go on with the next chunk's buffers."""
        import torch
        p = self._pipe
        xs = p["xstream"]
        for o in p["streams"]:
            e = torch.cuda.Event()
            e.record(o)
            xs.wait_event(e)
        with torch.cuda.stream(xs):
            w = self._dist.all_gather_into_tensor(p["gathered"][c].view(-1), p["local"][c], group=self.group, async_op=True)
            w.wait()    # orders this stream behind the collective; does not block the host
            if self.rank == self.dst:
                p["host"][c].copy_(p["gathered"][c], non_blocking=True)
                done = torch.cuda.Event()
                done.record(xs)
                p["done"][c] = done          # chunk c is home when this fires (collect merges it while later chunks still run)
        p["sent"] = c + 1

    def enqueue(self, query_ptr: int, d: int) -> int:
        """Enqueue one search (device pointer to d f32) without synchronising; returns its ticket.
        Consecutive searches alternate streams: the next query's score kernel fills the CUs this
        query's small top-k kernels leave idle.  Every ``gather_every``-th one issues the exchange."""
        p = self._pipe
        i = p["n"]
        g = self.gather_every
        c, j = divmod(i, g)
        assert c < p["nchunks"], "ShardedIndex.open(capacity) exceeded"
        st = p["streams"][i % len(p["streams"])]
        base = (p["host"][c, 0] if not self._multi else p["local"][c]).data_ptr() + j * p["rec"]
        self.local.search_device(query_ptr, 1, d, p["k"], base, base + p["s_bytes"], st.cuda_stream)
        p["n"] = i + 1
        if self._multi and j == g - 1:
            self._send_chunk(c, st)
        return i

    def collect(self, first: int = 0) -> Optional[List[Tuple[np.ndarray, np.ndarray]]]:
        """Exchanges a partly filled last chunk, drains the streams, and merges tickets
        ``first`` .. on ``dst`` (host merge under the single-index total order).  Collective:
        every rank calls it at the same point."""
        import torch
        p = self._pipe
        g = self.gather_every
        if self._multi and p["n"] > p["sent"] * g:
            last = p["n"] - 1
            self._send_chunk(last // g, p["streams"][last % len(p["streams"])])
        if self.rank != self.dst or not self._multi:
            torch.cuda.synchronize(self.device)
        if self.rank != self.dst:
            return None
        k, rec, s_bytes = p["k"], p["rec"], p["s_bytes"]
        count = min(k, self.n_total)
        out = []
        if not self._multi:
            for i in range(first, p["n"]):
                c, j = divmod(i, g)
                sc, rw = unpack_records(p["host"][c].numpy()[:, j * rec:(j + 1) * rec], self.world, k)
                out.append((sc[0, :count].copy(), rw[0, :count].copy()))
            return out
        # chunk by chunk, in the order they come home: chunk c's records are merged (all its steps in one call)
        # while the GPU is still searching for the later ones -- the host merge of 8 x 100 candidates per step
        # (10 us; 56 us with the lexsort of rounds 1-3, serial, AFTER the last kernel: a third of an 8-GPU
        # strong-scaling step) leaves the timed path except for the last chunk's
        for c in range(first // g, (p["n"] + g - 1) // g):
            p["done"][c].synchronize()
            j0, j1 = max(first - c * g, 0), min(p["n"] - c * g, g)
            buf = p["host"][c].numpy().reshape(self.world, g, rec)[:, j0:j1]
            sc = np.ascontiguousarray(buf[:, :, :k * 4]).view(np.float32).reshape(self.world, j1 - j0, k)
            rw = np.ascontiguousarray(buf[:, :, s_bytes:s_bytes + k * 8]).view(np.int64).reshape(self.world, j1 - j0, k)
            ms, mr = merge_topk_batch(sc, rw, count)
            out.extend((ms[j], mr[j]) for j in range(j1 - j0))
        torch.cuda.synchronize(self.device)
        return out

    @property
    def streams(self) -> int:
        return self._n_streams
