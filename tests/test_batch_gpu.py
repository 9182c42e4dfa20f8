"""GPU: batched search (16 queries per corpus pass, f32 MFMA) against the numpy
oracle, query by query.  A batch is by definition the per-query results in order
(the reference has no batched entry: it loops np.dot, src/svs/kb.py:1623)."""
import numpy as np
import pytest

from compare import assert_topk_parity
from oracle import svs_oracle as oracle
from synth import corpus_and_query

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,d,nq,k", [(20000, 1536, 2, 100), (20000, 1536, 16, 100), (20000, 1536, 17, 100),
                                      (9999, 768, 40, 50), (5000, 1024, 5, 5000), (4000, 128, 33, 10),
                                      (30001, 2304, 3, 100), (300, 1536, 20, 100)])
def test_batch_matches_oracle(gpu, n, d, nq, k):
    from svs_amd import DeviceIndex
    m, qs = corpus_and_query("gaussian", 1000 + n + nq, n, d, nq)
    idx = DeviceIndex(m)
    bs, br = idx.search_batch(qs, k)
    assert bs.shape == (nq, min(k, n)) and br.shape == bs.shape
    for qi, q in enumerate(qs):
        exp = oracle.cpu_search(m, q, k)
        assert_topk_parity(bs[qi], br[qi], [s for s, _ in exp], [i for _, i in exp],
                           oracle.cpu_scores_f64(m, q), label=f"{n}x{d} nq={nq} q{qi}")
    # batched and single-query kernels sum in different orders: same rows (bar near ties), scores within 2e-6
    for qi in (0, nq - 1):
        one = idx.search(qs[qi], k)
        assert np.max(np.abs(np.array([s for s, _ in one]) - bs[qi])) < 2e-6
    idx.release()


def test_batch_f32_accuracy_vs_f64(gpu):
    """The MFMA f32 chain is exact f32 (no reduced precision): error vs f64 stays
    at the 1e-7 level over d = 1536."""
    from svs_amd import DeviceIndex
    m, qs = corpus_and_query("gaussian", 77, 8192, 1536, 16)
    idx = DeviceIndex(m)
    bs, br = idx.search_batch(qs, 8192)   # full ranking: every score comes back
    for qi, q in enumerate(qs):
        truth = oracle.cpu_scores_f64(m, q)
        got = np.empty(8192); got[br[qi]] = bs[qi]
        assert np.max(np.abs(got - truth)) < 5e-7
    idx.release()


@pytest.mark.parametrize("kind,n,d,nq,k", [("gaussian", 150000, 1536, 16, 100), ("gaussian", 140001, 512, 40, 100),
                                           ("uniform", 131072, 256, 16, 10), ("gaussian", 135000, 128, 17, 256),
                                           ("gaussian", 200000, 768, 48, 100)])
def test_batch_f32_fused_topk(gpu, kind, n, d, nq, k):
    """f32, >= 16 queries over >= 131,072 rows: the batched kernels run with the fused
    top-k epilogue (no score matrix; thresholds from a prefix of the rows) -- the
    16-query streaming kernel up to 16 queries, the tiled kernel beyond."""
    from svs_amd import DeviceIndex
    m, qs = corpus_and_query(kind, 9000 + n + nq, n, d, nq)
    idx = DeviceIndex(m)
    bs, br = idx.search_batch(qs, k)
    assert bs.shape == (nq, k)
    for qi in range(0, nq, max(1, nq // 8)):
        exp = oracle.cpu_search(m, qs[qi], k)
        assert_topk_parity(bs[qi], br[qi], [s for s, _ in exp], [i for _, i in exp],
                           oracle.cpu_scores_f64(m, qs[qi]), label=f"f32 fused {kind} {n}x{d} nq={nq} q{qi}")
    # the materialised path (7 queries: below the fused minimum) gives the same answer bit for bit:
    # same kernel, same summation order
    ms, mr = idx.search_batch(qs[:7], k)
    if nq <= 16:
        assert np.array_equal(mr, br[:7]) and np.array_equal(ms, bs[:7])
    idx.release()


def test_batch_f32_fused_overflow_falls_back_exactly(gpu, first_rows_thresholds):
    """Adversarial row order (scores rise with the row index): every row passes the
    prefix threshold, the candidate lists overflow, and those queries are re-run
    through the materialised path -- still exact."""
    from svs_amd import DeviceIndex
    rng = np.random.default_rng(12)
    n, d, nq, k = 150000, 128, 16, 50
    u = rng.standard_normal(d); u /= np.linalg.norm(u)
    v = rng.standard_normal((n, d)); v -= np.outer(v @ u, u); v /= np.linalg.norm(v, axis=1, keepdims=True)
    c = np.linspace(0.05, 0.95, n)[:, None]
    m = (c * u[None, :] + np.sqrt(1 - c * c) * v).astype(np.float32)
    qs = rng.standard_normal((nq, d)); qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    qs[:4] = u
    qs = qs.astype(np.float32)
    idx = DeviceIndex(m)
    bs, br = idx.search_batch(qs, k)
    for qi in range(nq):
        exp = oracle.cpu_search(m, qs[qi], k)
        assert_topk_parity(bs[qi], br[qi], [s for s, _ in exp], [i for _, i in exp],
                           oracle.cpu_scores_f64(m, qs[qi]), label=f"f32 overflow q{qi}")
    idx.release()


def test_concurrent_mixed_batches_on_one_handle(gpu):
    """AsyncKB-style concurrency (reference src/svs/kb.py:1184-1190: searches run on executor
    threads outside the lock) with the batched paths in the mix: single-query GEMV, the
    16-query streaming kernel with the fused epilogue, the tiled kernel (fused), and the
    materialised path, all on ONE handle at once.  Every call must return exactly what it
    returns when it runs alone (per-call contexts: stream, scratch, candidate lists)."""
    import threading
    from svs_amd import DeviceIndex
    m, qs = corpus_and_query("gaussian", 31337, 150000, 512, 48)
    idx = DeviceIndex(m)
    jobs = [("single", qs[0:1]), ("b16 fused", qs[0:16]), ("b7 materialised", qs[16:23]),
            ("b40 tiled fused", qs[8:48]), ("b17", qs[3:20])]
    k = 64
    expected = {name: idx.search_batch(q, k) for name, q in jobs}
    errors = []

    def worker(t):
        try:
            for it in range(12):
                name, q = jobs[(t + it) % len(jobs)]
                s, r = idx.search_batch(q, k)
                es, er = expected[name]
                assert np.array_equal(r, er) and np.array_equal(s, es), f"thread {t} iteration {it}: {name}"
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    idx.release()
    assert not errors, errors
