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


# ---- wire format of one rank's local top-k (one query) -----------------------
# [k f32 scores | pad to 8 B | k i64 global rows]; entries past the shard's count
# carry row = -1.  Exchanged as raw bytes so ONE collective moves both arrays.
def record_layout(k: int) -> Tuple[int, int]:
    """(offset of the row array, record bytes)."""
    s_bytes = (k * 4 + 7) // 8 * 8
    return s_bytes, s_bytes + k * 8


def unpack_records(buf: np.ndarray, world: int, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """``buf``: uint8 (world, record_bytes) -> (scores f32 (world,k), rows i64 (world,k))."""
    s_bytes, rec = record_layout(k)
    buf = np.ascontiguousarray(buf, dtype=np.uint8).reshape(world, rec)
    sc = np.ascontiguousarray(buf[:, : k * 4]).view(np.float32).reshape(world, k)
    rw = np.ascontiguousarray(buf[:, s_bytes:]).view(np.int64).reshape(world, k)
    return sc, rw


def pack_record(scores: np.ndarray, rows: np.ndarray, k: int) -> np.ndarray:
    """Host-side packer (tests / CPU paths); the GPU path has the search kernel
    write both arrays straight into the record."""
    s_bytes, rec = record_layout(k)
    out = np.zeros(rec, dtype=np.uint8)
    sc = np.full(k, -np.inf, dtype=np.float32)
    rw = np.full(k, -1, dtype=np.int64)
    sc[: len(scores)] = scores
    rw[: len(rows)] = rows
    out[: k * 4] = sc.view(np.uint8)
    out[s_bytes:] = rw.view(np.uint8)
    return out


class ShardedIndex:
    """Distributed wrapper: ``local_search(queries (nq,d) f32, k) ->
    (scores (nq,c) f32, rows (nq,c) i64 GLOBAL)`` is the per-rank search (a
    ``DeviceIndex.search_batch`` built with ``row_offset = shard_bounds()[0]``);
    this class adds the gather + merge.  Results are returned on ``dst`` (None
    elsewhere)."""

    def __init__(self, local_search: Callable[[np.ndarray, int], Tuple[np.ndarray, np.ndarray]],
                 n_total: int, group=None, dst: int = 0, device=None):
        import torch.distributed as dist
        self._dist = dist
        self.local_search = local_search
        self.n_total = int(n_total)
        self.group = group
        self.dst = dst
        self.device = device
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def search_batch(self, queries: np.ndarray, n: int) -> Optional[Tuple[np.ndarray, np.ndarray]]:
        import torch
        assert isinstance(n, int)
        q = np.ascontiguousarray(queries, dtype=np.float32)
        nq = q.shape[0]
        k = max(n, 0)
        ls, lr = self.local_search(q, k)
        # fixed-size message: pad every shard's list to k (a shard may hold < k rows)
        ps = np.full((nq, k), -np.inf, dtype=np.float32)
        pr = np.full((nq, k), -1, dtype=np.int64)
        ps[:, : ls.shape[1]] = ls
        pr[:, : lr.shape[1]] = lr
        if self.world == 1:
            gs, gr = ps[None], pr[None]
        else:
            ts = torch.from_numpy(ps)
            tr = torch.from_numpy(pr)
            if self.device is not None:
                ts, tr = ts.to(self.device), tr.to(self.device)
            outs = [torch.empty_like(ts) for _ in range(self.world)] if self.rank == self.dst else None
            outr = [torch.empty_like(tr) for _ in range(self.world)] if self.rank == self.dst else None
            self._dist.gather(ts, outs, dst=self.dst, group=self.group)
            self._dist.gather(tr, outr, dst=self.dst, group=self.group)
            if self.rank != self.dst:
                return None
            gs = torch.stack(outs).cpu().numpy()
            gr = torch.stack(outr).cpu().numpy()
        count = min(k, self.n_total)
        out_s = np.empty((nq, count), dtype=np.float32)
        out_r = np.empty((nq, count), dtype=np.int64)
        for i in range(nq):
            s, r = merge_topk(gs[:, i, :], gr[:, i, :], count)
            out_s[i], out_r[i] = s, r
        return out_s, out_r

    def search(self, query_vec: np.ndarray, n: int) -> Optional[List[Tuple[float, int]]]:
        res = self.search_batch(np.asarray(query_vec, dtype=np.float32)[None, :], n)
        if res is None:
            return None
        return [(float(a), int(b)) for a, b in zip(res[0][0], res[1][0])]
