"""TEST DOUBLE for the CPU suite only: an object with DeviceIndex's surface whose
arithmetic is the numpy oracle.  It lets the host logic (KB mirror, matrix cache,
attach(), row sharding) be exercised in the build container, which has no GPU.
It is never importable from the product (tests/ is not a package dependency of
svs_amd), and GPU tests do not use it."""
import threading

import numpy as np

from oracle import svs_oracle as oracle


class OracleIndex:
    live = 0   # number of un-released instances (leak checks)
    _mu = threading.Lock()

    def __init__(self, matrix, device=0, row_offset=0, _shared=None):
        # state shared by every owner of the same "HBM copy": [matrix, dead-row mask]
        if _shared is not None:
            self._st = _shared
        else:
            m = np.ascontiguousarray(matrix, dtype=np.float32)
            assert m.ndim == 2
            self._st = [m, np.zeros(m.shape[0], dtype=bool)]
        self.device, self.row_offset = device, row_offset
        self._released = False
        with OracleIndex._mu:
            OracleIndex.live += 1

    def __len__(self):
        return self.n

    @property
    def _m(self):
        return self._st[0]

    @property
    def n(self):
        return self._st[0].shape[0]

    @property
    def d(self):
        return self._st[0].shape[1]

    @property
    def shape(self):
        return self._st[0].shape

    def share(self):
        self._check()
        return OracleIndex(None, self.device, self.row_offset, _shared=self._st)

    def append(self, rows):
        self._check()
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        self._st[0] = np.vstack([self._st[0], rows])
        self._st[1] = np.concatenate([self._st[1], np.zeros(rows.shape[0], dtype=bool)])

    def mask_rows(self, rows):
        self._check()
        self._st[1][np.asarray(rows, dtype=np.int64) - self.row_offset] = True

    def _scores(self, q):
        s = oracle.cpu_scores(self._m, np.asarray(q, dtype=np.float32)).copy()
        s[self._st[1]] = -np.inf
        return s

    def _check(self):
        if self._released:
            raise RuntimeError("DeviceIndex has been released")

    def release(self):
        if not self._released:
            self._released = True
            with OracleIndex._mu:
                OracleIndex.live -= 1

    close = release

    def scores(self, q):
        self._check()
        return oracle.cpu_scores(self._m, np.asarray(q, dtype=np.float32))

    def search(self, q, n):
        self._check()
        assert isinstance(n, int)
        q = np.asarray(q, dtype=np.float32)
        if q.ndim != 1 or q.shape[0] != self.d or self.n == 0:
            raise ValueError(f"shapes {self.shape} and {q.shape} not aligned")
        n = min(n, int((~self._st[1]).sum()))
        return [(s, i + self.row_offset) for s, i in oracle.total_order_top_k(self._scores(q), n)]

    def top_pairs(self, n):
        self._check()
        g = np.dot(self._m, self._m.T)
        g[self._st[1], :] = -np.inf
        g[:, self._st[1]] = -np.inf
        live = int((~self._st[1]).sum())
        return oracle.cpu_top_pairs(g, min(n, live * (live - 1) // 2))

    def search_batch(self, queries, n):
        self._check()
        q = np.ascontiguousarray(queries, dtype=np.float32)
        c = min(max(n, 0), int((~self._st[1]).sum()))
        s = np.empty((q.shape[0], c), dtype=np.float32)
        r = np.empty((q.shape[0], c), dtype=np.int64)
        for i, qq in enumerate(q):
            res = self.search(qq, n)
            s[i] = [a for a, _ in res]
            r[i] = [b for _, b in res]
        return s, r
