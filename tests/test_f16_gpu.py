"""GPU: f16-resident corpus (SVS_DTYPE_F16).  Oracle = numpy's f32 path on the
dequantised (half-rounded, upcast to f32) corpus and query -- SURVEY.md 8(d)."""
import numpy as np
import pytest

from compare import assert_topk_parity
from oracle import svs_oracle as oracle
from synth import corpus_and_query

pytestmark = pytest.mark.gpu


def _deq(x):
    return x.astype(np.float16).astype(np.float32)


@pytest.mark.parametrize("n,d,k", [(30000, 1536, 100), (9000, 3072, 100), (5000, 512, 10), (7001, 1024, 64),
                                   (4000, 768, 100), (3000, 100, 7), (600, 3, 5), (50, 1537, 100)])
def test_f16_single_query(gpu, n, d, k):
    from svs_amd import DeviceIndex
    m, qs = corpus_and_query("gaussian", 500 + n, n, d, 3)
    idx = DeviceIndex(m, dtype="f16")
    assert idx.dtype == "f16" and idx.ld % 8 == 0 and idx.ld >= d and idx.hbm_bytes == n * idx.ld * 2
    md = _deq(m)
    for q in qs:
        qd = _deq(q)
        got = idx.search(q, k)
        exp = oracle.cpu_search(md, qd, k)
        assert_topk_parity([s for s, _ in got], [i for _, i in got], [s for s, _ in exp], [i for _, i in exp],
                           oracle.cpu_scores_f64(md, qd), label=f"f16 {n}x{d}")
    sc = idx.scores(qs[0])
    assert np.max(np.abs(sc - oracle.cpu_scores_f64(md, _deq(qs[0])))) < 5e-7
    # what the index holds IS numpy's half rounding of the corpus / query
    assert np.array_equal(idx.stored_rows(0, min(n, 64)), md[:64])
    assert np.array_equal(idx.stored_query(qs[0]), _deq(qs[0]))
    idx.release()


@pytest.mark.parametrize("n,d,nq,k", [(20000, 1536, 32, 100), (20000, 1536, 33, 100), (9000, 3072, 20, 50),
                                      (5000, 768, 7, 100), (12000, 256, 70, 10)])
def test_f16_batch_mfma(gpu, n, d, nq, k):
    from svs_amd import DeviceIndex
    m, qs = corpus_and_query("gaussian", 900 + n + nq, n, d, nq)
    idx = DeviceIndex(m, dtype="f16")
    md = _deq(m)
    bs, br = idx.search_batch(qs, k)
    for qi, q in enumerate(qs):
        qd = _deq(q)
        exp = oracle.cpu_search(md, qd, k)
        assert_topk_parity(bs[qi], br[qi], [s for s, _ in exp], [i for _, i in exp],
                           oracle.cpu_scores_f64(md, qd), label=f"f16 batch {n}x{d} q{qi}")
    idx.release()


def test_f16_recall_vs_f32(gpu):
    """Rounding the corpus to half barely moves the ranking: recall@100 vs the f32 truth."""
    from svs_amd import DeviceIndex
    m, qs = corpus_and_query("gaussian", 4242, 50000, 1536, 8)
    idx = DeviceIndex(m, dtype="f16")
    rec = []
    for q in qs:
        got = {i for _, i in idx.search(q, 100)}
        exp = {i for _, i in oracle.cpu_search(m, q, 100)}
        rec.append(len(got & exp) / 100)
    idx.release()
    assert min(rec) >= 0.95, rec


@pytest.mark.parametrize("n,d,nq,k", [(140000, 768, 64, 100), (131072, 256, 100, 10), (150001, 512, 300, 100),
                                      (40000, 1536, 64, 100), (140000, 768, 17, 100), (135000, 1536, 33, 50)])
def test_f16_large_batch_fused_topk(gpu, n, d, nq, k):
    """nq >= 16 over >= 131,072 rows takes the batched kernels with the fused top-k
    epilogue (the score matrix is never materialised; thresholds come from a
    16,384-row prefix); smaller corpora take the materialised path."""
    from svs_amd import DeviceIndex
    m, qs = corpus_and_query("gaussian", 7000 + n, n, d, nq)
    idx = DeviceIndex(m, dtype="f16")
    md = _deq(m)
    bs, br = idx.search_batch(qs, k)
    for qi in range(0, nq, max(1, nq // 16)):
        qd = _deq(qs[qi])
        exp = oracle.cpu_search(md, qd, k)
        assert_topk_parity(bs[qi], br[qi], [s for s, _ in exp], [i for _, i in exp],
                           oracle.cpu_scores_f64(md, qd), label=f"f16 fused {n}x{d} q{qi}")
    # same results as the materialised path (queries one at a time), up to near ties
    for qi in (0, nq - 1):
        one = idx.search(qs[qi], k)
        assert np.max(np.abs(np.array([s for s, _ in one]) - bs[qi])) < 2e-6
    idx.release()


def test_f16_fused_overflow_falls_back_exactly(gpu, first_rows_thresholds):
    """Adversarial row order: scores rise with the row index, so every row passes
    the streaming cut and the candidate lists overflow; those queries must be
    re-run through the materialised path and still be exact."""
    from svs_amd import DeviceIndex
    rng = np.random.default_rng(11)
    n, d, nq, k = 150000, 64, 70, 50
    u = rng.standard_normal(d); u /= np.linalg.norm(u)
    v = rng.standard_normal((n, d)); v -= np.outer(v @ u, u); v /= np.linalg.norm(v, axis=1, keepdims=True)
    c = np.linspace(0.05, 0.95, n)[:, None]                 # cosine to u grows with the row
    m = (c * u[None, :] + np.sqrt(1 - c * c) * v).astype(np.float32)
    qs = rng.standard_normal((nq, d)); qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    qs[:8] = u                                              # 8 adversarial queries
    qs = qs.astype(np.float32)
    idx = DeviceIndex(m, dtype="f16")
    md = _deq(m)
    bs, br = idx.search_batch(qs, k)
    for qi in list(range(10)) + [nq - 1]:
        qd = _deq(qs[qi])
        exp = oracle.cpu_search(md, qd, k)
        assert_topk_parity(bs[qi], br[qi], [s for s, _ in exp], [i for _, i in exp],
                           oracle.cpu_scores_f64(md, qd), label=f"overflow q{qi}")
    idx.release()


def test_f16_a_whole_batch_of_overflowing_queries_is_rerun_in_batches(gpu, first_rows_thresholds):
    """A corpus SORTED by similarity to the queries (every query's candidate list overflows): the host entry re-runs
    them through the materialised path, 64 per pass, not one single-query search each -- exact, and the call stays
    within a small multiple of an ordinary batch's time (rounds 2-3: one 0.2-0.9 ms search per query)."""
    import ctypes as C
    import time
    from svs_amd import DeviceIndex, _native
    rng = np.random.default_rng(12)
    n, d, nq, k = 150000, 64, 200, 50
    u = rng.standard_normal(d); u /= np.linalg.norm(u)
    v = rng.standard_normal((n, d)); v -= np.outer(v @ u, u); v /= np.linalg.norm(v, axis=1, keepdims=True)
    c = np.linspace(0.05, 0.95, n)[:, None]
    m = (c * u[None, :] + np.sqrt(1 - c * c) * v).astype(np.float32)
    qs = u[None, :] + 0.05 * rng.standard_normal((nq, d))      # every query close to u
    qs = (qs / np.linalg.norm(qs, axis=1, keepdims=True)).astype(np.float32)
    idx = DeviceIndex(m, dtype="f16")
    md = _deq(m)
    idx.search_batch(qs, k)
    t0 = time.perf_counter()
    bs, br = idx.search_batch(qs, k)
    dt = time.perf_counter() - t0
    ph = (C.c_double * 6)()
    _native.load().svs_internal_host_phases(ph, 6)
    n_rerun = int(ph[5])
    assert n_rerun >= nq // 2, f"expected most of the {nq} queries to overflow and be re-run, got {n_rerun}"
    for qi in (0, 1, 63, 64, 65, 127, 128, nq - 1):
        qd = _deq(qs[qi])
        exp = oracle.cpu_search(md, qd, k)
        assert_topk_parity(bs[qi], br[qi], [s for s, _ in exp], [i for _, i in exp],
                           oracle.cpu_scores_f64(md, qd), label=f"batched re-run q{qi}")
    # an ordinary batch of the same shape, for scale (random queries: nothing overflows)
    qr = rng.standard_normal((nq, d)).astype(np.float32); qr /= np.linalg.norm(qr, axis=1, keepdims=True)
    idx.search_batch(qr, k)
    t0 = time.perf_counter()
    idx.search_batch(qr, k)
    dt_plain = time.perf_counter() - t0
    _native.load().svs_internal_host_phases(ph, 6)
    assert ph[5] == 0
    print(f"all-overflow batch {dt * 1e3:.2f} ms ({n_rerun} re-run) vs ordinary batch {dt_plain * 1e3:.2f} ms")
    assert dt < 25 * dt_plain + 5e-3, (dt, dt_plain)
    idx.release()


@pytest.mark.parametrize("dtype,nq", [("f16", 200), ("f16", 16), ("f32", 16), ("fp8", 64)])
def test_sorted_corpus_stays_on_the_fused_path(gpu, dtype, nq):
    """The thresholds come from a sample spread over the whole corpus (blocks of 256 rows every n / 64 rows), so a
    corpus SORTED by similarity to the queries is no worse than a shuffled one: nothing overflows, nothing is re-run,
    the answers are exact -- also with the very best rows tombstoned, some of them inside the sample's blocks."""
    import ctypes as C
    from svs_amd import DeviceIndex, _native
    rng = np.random.default_rng(21)
    n, d, k = 150000, 128, 50
    u = rng.standard_normal(d); u /= np.linalg.norm(u)
    v = rng.standard_normal((n, d)); v -= np.outer(v @ u, u); v /= np.linalg.norm(v, axis=1, keepdims=True)
    c = np.linspace(0.05, 0.95, n)[:, None]
    m = (c * u[None, :] + np.sqrt(1 - c * c) * v).astype(np.float32)
    qs = u[None, :] + 0.05 * rng.standard_normal((nq, d))
    qs = (qs / np.linalg.norm(qs, axis=1, keepdims=True)).astype(np.float32)
    idx = DeviceIndex(m, dtype=dtype)
    md = m if dtype == "f32" else idx.stored_rows()
    ph = (C.c_double * 6)()
    dead = np.array([], dtype=np.int64)
    for round_ in range(2):
        bs, br = idx.search_batch(qs, k)
        _native.load().svs_internal_host_phases(ph, 6)
        assert ph[5] == 0, f"{int(ph[5])} queries were re-run: the thresholds cut too little"
        for qi in (0, 1, nq // 2, nq - 1):
            qd = qs[qi] if dtype == "f32" else idx.stored_query(qs[qi])
            sc = oracle.cpu_scores_f64(md, qd)
            if len(dead):
                sc = sc.copy(); sc[dead] = -np.inf
            order = np.lexsort((np.arange(n), sc))[::-1][:k]
            assert not set(br[qi].tolist()) & set(dead.tolist())
            assert_topk_parity(bs[qi], br[qi], sc[order].astype(np.float32), order.tolist(), sc, label=f"sorted corpus {dtype} q{qi} round {round_}",
                               score_atol=1e-5 if dtype != "fp8" else 3e-4)
        # second round: the best rows are tombstoned -- the last 300 of the corpus and the whole LAST BLOCK of the sample
        # (64 blocks of 256 rows at this size, one every n // 64 rows): the thresholds must come from live rows only
        stride = n // 64
        dead = np.unique(np.r_[np.arange(n - 300, n), np.arange(63 * stride, 63 * stride + 256), np.arange(62 * stride + 100, 62 * stride + 130)])
        idx.mask_rows(dead)
    idx.release()


@pytest.mark.parametrize("n,d,nq,k", [(20000, 1536, 2, 100), (20000, 1536, 16, 100), (15000, 1280, 9, 50),
                                      (15000, 1664, 12, 50), (15000, 1408, 16, 50), (6000, 4608, 5, 20),
                                      (140000, 512, 16, 100), (133000, 1024, 16, 256)])
def test_f16_small_batches_streaming_kernel(gpu, n, d, nq, k):
    """Up to 16 queries over an f16 corpus whose rows are whole pairs of 256-byte steps take the
    streaming kernel (v_mfma_f32_4x4x4_16b_f16, whole-line loads); other dimensions (1664 and
    1408 halves here) the tiled kernel; 16 queries over >= 131,072 rows run it with the fused
    top-k epilogue.  All against numpy on the half-rounded corpus and queries."""
    from svs_amd import DeviceIndex
    m, qs = corpus_and_query("gaussian", 8100 + n + d + nq, n, d, nq)
    idx = DeviceIndex(m, dtype="f16")
    md = _deq(m)
    bs, br = idx.search_batch(qs, k)
    assert bs.shape == (nq, k)
    for qi in range(nq):
        qd = _deq(qs[qi])
        exp = oracle.cpu_search(md, qd, k)
        assert_topk_parity(bs[qi], br[qi], [s for s, _ in exp], [i for _, i in exp],
                           oracle.cpu_scores_f64(md, qd), label=f"f16 stream {n}x{d} nq={nq} q{qi}")
    idx.release()
