"""SearchCoalescer (svs_amd/coalesce.py): concurrent single-query searches -- the executor threads
of AsyncKB.retrieve, reference src/svs/kb.py:1184-1190 -- share corpus passes.  CPU part: the queueing
logic against a slow fake index; GPU part: real searches from 24 threads."""
import threading
import time

import numpy as np
import pytest

from svs_amd.coalesce import SearchCoalescer


class _SlowIndex:
    """numpy index that takes 20 ms per pass, whatever the batch size (like the GPU's score stage)."""

    def __init__(self, m):
        self.m, self.d = m, m.shape[1]
        self.single = self.batched = 0
        self.batch_sizes = []

    def _topk(self, q, n):
        x = self.m @ q
        order = sorted(((float(s), i) for i, s in enumerate(x)), reverse=True)
        return order[: min(n, len(order))]

    def search(self, q, n):
        assert isinstance(n, int)
        if np.asarray(q).shape != (self.d,):
            raise ValueError("shapes not aligned")
        time.sleep(0.02)
        self.single += 1
        return self._topk(np.asarray(q, dtype=np.float32), n)

    def search_batch(self, qs, n):
        time.sleep(0.02)
        self.batched += 1
        self.batch_sizes.append(len(qs))
        res = [self._topk(q, n) for q in qs]
        c = len(res[0])
        return (np.array([[s for s, _ in r] for r in res], dtype=np.float32).reshape(len(qs), c),
                np.array([[i for _, i in r] for r in res], dtype=np.int64).reshape(len(qs), c))


def test_coalescer_batches_concurrent_callers_and_keeps_answers():
    rng = np.random.default_rng(0)
    m = rng.standard_normal((500, 16)).astype(np.float32)
    qs = rng.standard_normal((40, 16)).astype(np.float32)
    idx = _SlowIndex(m)
    want = [idx._topk(q, 7 + i % 5) for i, q in enumerate(qs)]
    co = SearchCoalescer(max_batch=16)
    got = [None] * len(qs)

    def worker(i):
        got[i] = co.search(idx, qs[i], 7 + i % 5)        # different n per caller: served from one k = max pass

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(len(qs))]
    t0 = time.time()
    [t.start() for t in ts]
    [t.join() for t in ts]
    dt = time.time() - t0
    for i in range(len(qs)):
        assert [r for _, r in got[i]] == [r for _, r in want[i]]
        assert np.allclose([s for s, _ in got[i]], [s for s, _ in want[i]], atol=1e-6)
    assert co.queries == len(qs) and co.batches < len(qs) / 3, (co.batches, idx.batch_sizes)
    assert max(idx.batch_sizes) <= 16
    assert dt < 0.02 * len(qs) / 2                        # 40 solo passes would take 0.8 s
    assert not co._busy and not co._pending


def test_coalescer_alone_is_the_plain_search_and_errors_stay_with_their_caller():
    rng = np.random.default_rng(1)
    idx = _SlowIndex(rng.standard_normal((50, 8)).astype(np.float32))
    co = SearchCoalescer()
    q = rng.standard_normal(8).astype(np.float32)
    assert co.search(idx, q, 5) == idx._topk(q, 5) and idx.batched == 0 and idx.single == 1
    assert co.search(idx, q, 0) == [] or idx.single == 2          # n <= 0 goes straight to the index
    with pytest.raises(ValueError):
        co.search(idx, np.zeros(9, dtype=np.float32), 5)           # wrong dimension: the caller's own error
    with pytest.raises(AssertionError):
        co.search(idx, q, np.int64(5))
    assert not co._busy and not co._pending

    class _Broken(_SlowIndex):
        def search_batch(self, qs, n):
            time.sleep(0.02)
            raise RuntimeError("device lost")

    bad = _Broken(idx.m)
    errs = []

    def worker():
        try:
            co2.search(bad, q, 3)
        except RuntimeError as e:
            errs.append(str(e))
        except BaseException as e:   # noqa: BLE001
            errs.append(repr(e))

    co2 = SearchCoalescer()
    ts = [threading.Thread(target=worker) for _ in range(6)]
    [t.start() for t in ts]
    [t.join(timeout=10) for t in ts]
    assert not any(t.is_alive() for t in ts), "a waiter was never released"
    assert errs.count("device lost") >= 1 and not co2._busy and not co2._pending


@pytest.mark.gpu
def test_coalesced_retrieves_on_the_gpu(gpu):
    """24 threads through DeviceEmbeddingsMatrix.search (what KB.retrieve / AsyncKB.retrieve call): rows
    equal to the solo searches', scores within the f32 summation noise, fewer corpus passes than searches."""
    from svs_amd.matrix import DeviceEmbeddingsMatrix
    from synth import corpus_and_query
    m, qs = corpus_and_query("gaussian", 31337, 300000, 1536, 96)
    ids = np.arange(1000, 1000 + len(m), dtype=np.int64)
    mat = DeviceEmbeddingsMatrix(builder=lambda db: (m, ids), keep_host_matrix=False)
    mat.get_sync(None)
    mat.coalesce = False
    want = [mat.search(q, 50) for q in qs]
    mat.coalesce = True
    got = [None] * len(qs)

    def worker(t):
        for i in range(t, len(qs), 24):
            got[i] = mat.search(qs[i], 50)

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(24)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for i in range(len(qs)):
        assert [e for _, e in got[i]] == [e for _, e in want[i]], i
        assert np.max(np.abs(np.array([s for s, _ in got[i]]) - np.array([s for s, _ in want[i]]))) <= 1e-5
    passes, answered = mat.index.coalesce_stats()           # a DeviceIndex coalesces inside the library
    assert answered == len(qs) and passes < len(qs), (answered, passes)
    print(f"{answered} searches in {passes} corpus passes")
    # the Python coalescer (what MultiDeviceIndex and other index types get) on the same index
    from svs_amd.coalesce import SearchCoalescer
    mat.index.set_coalesce(False)
    co = SearchCoalescer()
    idx = mat.index

    def worker2(t):
        for i in range(t, len(qs), 24):
            got[i] = [(s, int(ids[r])) for s, r in co.search(idx, qs[i], 50)]

    got = [None] * len(qs)
    ts = [threading.Thread(target=worker2, args=(t,)) for t in range(24)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for i in range(len(qs)):
        assert [e for _, e in got[i]] == [e for _, e in want[i]], i
    assert co.queries == len(qs) and co.batches < len(qs)
    mat.invalidate()


@pytest.mark.gpu
def test_native_coalescing_errors_and_mixed_k(gpu):
    """svs_index_set_coalesce: different k per caller, wrong dimension and k = 0 alongside, 32 threads."""
    from svs_amd import DeviceIndex
    from synth import corpus_and_query
    m, qs = corpus_and_query("gaussian", 424242, 200000, 768, 64)
    idx = DeviceIndex(m)
    want = [idx.search(q, 5 + i % 90) for i, q in enumerate(qs)]
    idx.set_coalesce(True)
    got, errs = [None] * len(qs), []

    def worker(t):
        for rep in range(3):
            for i in range(t, len(qs), 32):
                got[i] = idx.search(qs[i], 5 + i % 90)
            try:
                idx.search(np.zeros(769, dtype=np.float32), 3)
            except ValueError:
                errs.append(t)
            assert idx.search(qs[t], 0) == []

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(32)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert len(errs) == 32 * 3
    for i in range(len(qs)):
        assert [r for _, r in got[i]] == [r for _, r in want[i]], i
        assert np.max(np.abs(np.array([s for s, _ in got[i]]) - np.array([s for s, _ in want[i]]))) <= 1e-5
    passes, answered = idx.coalesce_stats()
    assert answered == 3 * len(qs) and passes < answered
    idx.release()
