"""GPU: the C ABI's multi-device entry (svs_multi_*, include/svs_amd.h; SURVEY.md 8(b) sketch
`devices, ndev`, 8(e) row sharding) against ONE index over the whole corpus: scores, rows and
order must be identical for any shard count.  The box has one GPU, so the shards share device 0
(the library allows a device to be listed more than once)."""
import numpy as np
import pytest

from synth import corpus_and_query

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,d,g,dtype", [(10548, 1536, 3, "f32"), (50000, 256, 8, "f32"), (7, 64, 3, "f32"),
                                         (20000, 768, 2, "f16"), (150000, 512, 4, "fp8")])
def test_native_multi_equals_single_index(gpu, n, d, g, dtype):
    from svs_amd import DeviceIndex
    from svs_amd.multi import NativeMultiIndex
    m, qs = corpus_and_query("gaussian", 4000 + n, n, d, 40)
    m[n // 2] = m[3]                                   # an exact tie across shards
    one = DeviceIndex(m, dtype=dtype)
    multi = NativeMultiIndex(m, devices=[0] * g, dtype=dtype)
    assert multi.shape == (n, d)
    for k in (1, 100, n + 5):
        assert multi.search(qs[0], k) == one.search(qs[0], k)
    s1, r1 = one.search_batch(qs, 50)
    s2, r2 = multi.search_batch(qs, 50)
    assert np.array_equal(r1, r2) and np.array_equal(s1, s2)
    assert multi.search(qs[0], 0) == [] and multi.search(qs[0], -3) == []
    with pytest.raises(ValueError):
        multi.search(np.zeros(d + 1, dtype=np.float32), 5)
    with pytest.raises(AssertionError):
        multi.search(qs[0], np.int64(5))
    # tombstones go to the shard that holds the row
    top = [r for _, r in one.search(qs[1], 10)]
    dead = [top[0], top[3], n - 1, 0]
    one.mask_rows(dead)
    multi.mask_rows(dead)
    a, b = one.search(qs[1], 20), multi.search(qs[1], 20)
    assert a == b and not (set(dead) & {r for _, r in b})
    assert len(multi.search(qs[1], n)) == n - len(set(dead))
    one.release()
    multi.release()


def test_native_multi_more_shards_than_rows_and_empty(gpu):
    from svs_amd.multi import NativeMultiIndex
    m, qs = corpus_and_query("gaussian", 77, 3, 16, 1)
    multi = NativeMultiIndex(m, devices=[0] * 5)      # two shards hold nothing
    got = multi.search(qs[0], 10)
    exp = sorted(((float(np.dot(m[i], qs[0])), i) for i in range(3)), reverse=True)
    assert [r for _, r in got] == [i for _, i in exp]
    multi.release()
    empty = NativeMultiIndex(np.zeros((0, 0), dtype=np.float32), devices=[0, 0])
    with pytest.raises(ValueError):                    # numpy: shapes (0,0) and (16,) not aligned
        empty.search(qs[0], 5)
    empty.release()


def test_native_multi_concurrent_callers(gpu):
    """svs_multi_search is re-entrant: callers queue their jobs on the shard workers."""
    import threading
    from svs_amd import DeviceIndex
    from svs_amd.multi import NativeMultiIndex
    m, qs = corpus_and_query("gaussian", 99, 60000, 384, 24)
    one = DeviceIndex(m)
    exp = [one.search(q, 30) for q in qs]
    multi = NativeMultiIndex(m, devices=[0, 0, 0])
    bad = []

    def worker(t):
        for rep in range(6):
            for qi in range(t, len(qs), 4):
                if multi.search(qs[qi], 30) != exp[qi]:
                    bad.append((t, qi))

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad
    # the same with coalescing in front of the shards: rows as before, scores within the summation noise
    multi.set_coalesce(True)
    bad2 = []

    def worker2(t):
        for rep in range(6):
            for qi in range(t, len(qs), 12):
                got = multi.search(qs[qi], 30)
                if [r for _, r in got] != [r for _, r in exp[qi]] or max(abs(a - b) for (a, _), (b, _) in zip(got, exp[qi])) > 1e-5:
                    bad2.append((t, qi))

    ts = [threading.Thread(target=worker2, args=(t,)) for t in range(12)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad2
    passes, answered = multi.coalesce_stats()
    assert answered == 6 * len(qs) and passes < answered
    one.release()
    multi.release()
