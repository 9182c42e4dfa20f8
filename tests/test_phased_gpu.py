"""GPU: the four-phase 256 x 256 MFMA kernel (svs_amd/csrc/gemm_phased.h; batches of more than 128
queries over f16 / fp8 corpora -- BASELINE.json configs[2] and configs[4]) against
  * the numpy oracle on the stored (rounded / quantised) corpus, and
  * the round-1 kernel it replaces (gemm_tiled_kernel<256, ., ., 256>, kept behind
    svs_index_set_variant(2)): same arithmetic in the same k order, so BIT-identical scores and rows.
The second check is also the race screen for the LDS-DMA ring (reads are placed by the
vmcnt / barrier counts; an early read shows up as a rare wrong tile): every shape is searched
several times, and the repetitions must agree bit for bit."""
import numpy as np
import pytest

from compare import assert_topk_parity
from oracle import svs_oracle as oracle
from synth import corpus_and_query

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,n,d,nq,k", [
    ("f16", 140000, 1536, 300, 100),   # fused top-k, two query tiles (one partial), last row tile partial
    ("f16", 131072 + 77, 256, 129, 50),  # 4 k-tiles per row, one query tile with 127 padded queries
    ("f16", 9000, 128, 256, 100),      # 2 k-tiles per row (the minimum), materialised scores (n < 131,072)
    ("f16", 300, 1536, 200, 300),      # fewer rows than one tile, k == n
    ("f16", 20000, 768, 1024, 10),     # 4 query tiles share every row tile
    ("fp8", 140000, 3072, 256, 100),   # configs[4]'s row length
    ("fp8", 10000, 256, 130, 20),      # 2 k-tiles per row
    ("fp8", 135000, 1536, 513, 64),    # three query tiles, the last with one query
    # few survivors per tile (small k): the epilogue's grouped compare-and-branch path (PG_SPARSE_MAX)
    ("f16", 140000, 1536, 300, 8),
    ("fp8", 140000, 3072, 256, 5),
])
def test_phased_equals_round1_kernel_and_oracle(gpu, dtype, n, d, nq, k):
    from svs_amd import DeviceIndex
    m, qs = corpus_and_query("gaussian", 900 + n % 1000 + nq, n, d, nq)
    idx = DeviceIndex(m, dtype=dtype)
    s0, r0 = idx.search_batch(qs, k)
    for rep in range(4):                 # race screen: repetitions agree bit for bit
        s1, r1 = idx.search_batch(qs, k)
        assert np.array_equal(r0, r1) and np.array_equal(s0, s1), f"run {rep} differs"
    idx.set_variant(2)                   # the round-1 256 x 256 kernel
    s2, r2 = idx.search_batch(qs, k)
    idx.set_variant(0)
    assert np.array_equal(r0, r2) and np.array_equal(s0, s2), "phased kernel != gemm_tiled kernel"
    md = idx.stored_rows()
    for qi in sorted({0, 1, nq // 2, 255 % nq, nq - 1}):
        qd = idx.stored_query(qs[qi])
        exp = oracle.cpu_search(md, qd, k)
        assert_topk_parity(s0[qi], r0[qi], [s for s, _ in exp], [i for _, i in exp],
                           oracle.cpu_scores_f64(md, qd), label=f"phased {dtype} {n}x{d} q{qi}")
    idx.release()


@pytest.mark.parametrize("dtype,n,d,nq,k", [
    ("f16", 140000, 1536, 128, 100),   # one full 128-query tile, fused top-k, last row tile partial
    ("f16", 140000, 1536, 65, 100),    # 63 padded queries (+inf thresholds in the sweeps)
    ("f16", 9000, 128, 100, 50),       # 2 k-tiles per row, materialised scores
    ("fp8", 140000, 3072, 128, 100),   # configs[4]'s row length
    ("fp8", 135000, 1536, 97, 7),      # few survivors per tile: the grouped epilogue path with padded queries
    ("fp8", 10000, 256, 66, 20),
])
def test_phased_128_query_tiles_equal_tiled_kernel_and_oracle(gpu, dtype, n, d, nq, k):
    """Panels of 65 .. 128 queries (the batches a coalescer forms on a reduced-precision index, reference
    src/svs/kb.py:1184-1190): gemm_phased_kernel<.., QT = 128> against the tiled kernel it replaces
    (svs_index_set_variant(2): bit-identical scores and rows) and the numpy oracle on the stored corpus."""
    from svs_amd import DeviceIndex
    m, qs = corpus_and_query("gaussian", 1700 + n % 1000 + nq, n, d, nq)
    idx = DeviceIndex(m, dtype=dtype)
    s0, r0 = idx.search_batch(qs, k)
    for rep in range(4):                 # race screen of the LDS-DMA ring at the new counted waits
        s1, r1 = idx.search_batch(qs, k)
        assert np.array_equal(r0, r1) and np.array_equal(s0, s1), f"run {rep} differs"
    idx.set_variant(2)
    s2, r2 = idx.search_batch(qs, k)
    idx.set_variant(0)
    assert np.array_equal(r0, r2) and np.array_equal(s0, s2), "128-query phased kernel != gemm_tiled kernel"
    md = idx.stored_rows()
    for qi in sorted({0, 1, nq // 2, 63, 64 % nq, nq - 1}):
        qd = idx.stored_query(qs[qi])
        exp = oracle.cpu_search(md, qd, k)
        assert_topk_parity(s0[qi], r0[qi], [s for s, _ in exp], [i for _, i in exp],
                           oracle.cpu_scores_f64(md, qd), label=f"phased-128 {dtype} {n}x{d} q{qi}")
    idx.release()


@pytest.mark.parametrize("dtype", ["f16", "fp8"])
@pytest.mark.parametrize("nq", [17, 64])
def test_small_panels_on_reduced_precision_match_oracle(gpu, dtype, nq):
    """17 and 64 queries over f16 / fp8 (the tiled MFMA kernels at 32- and 64-query tiles, fused top-k):
    oracle parity on the stored corpus, and batch == the per-query searches up to near ties."""
    from svs_amd import DeviceIndex
    n, d, k = 140000, 1536, 100
    m, qs = corpus_and_query("gaussian", 2100 + nq, n, d, nq)
    idx = DeviceIndex(m, dtype=dtype)
    s0, r0 = idx.search_batch(qs, k)
    md = idx.stored_rows()
    for qi in sorted({0, nq // 2, nq - 1}):
        qd = idx.stored_query(qs[qi])
        exp = oracle.cpu_search(md, qd, k)
        assert_topk_parity(s0[qi], r0[qi], [s for s, _ in exp], [i for _, i in exp],
                           oracle.cpu_scores_f64(md, qd), label=f"{dtype} panel of {nq} q{qi}")
    idx.release()


def test_phased_odd_ktile_count_falls_back(gpu):
    """Rows with an odd number of 128-byte k-tiles (d = 192 halves = 3 tiles) are not the phased
    kernel's: the tiled kernel serves them, results as the oracle's."""
    from svs_amd import DeviceIndex
    m, qs = corpus_and_query("gaussian", 5, 6000, 192, 200)
    idx = DeviceIndex(m, dtype="f16")
    s, r = idx.search_batch(qs, 20)
    md = idx.stored_rows()
    for qi in (0, 199):
        qd = idx.stored_query(qs[qi])
        exp = oracle.cpu_search(md, qd, 20)
        assert_topk_parity(s[qi], r[qi], [x for x, _ in exp], [i for _, i in exp], oracle.cpu_scores_f64(md, qd), label="odd k-tiles")
    idx.release()


@pytest.mark.parametrize("dtype", ["f16", "fp8"])
def test_phased_clustered_rows_overflow_the_parking_lot(gpu, dtype):
    """A corpus ordered by topic: 300 consecutive rows are near-copies of one query (and 150 of
    another), so single waves of the fused epilogue see more survivors than their eighth of the
    parking lot holds (128).  The rest goes straight to the queries' global candidate lists; results
    must be the oracle's and, bit for bit, the round-1 kernel's."""
    from svs_amd import DeviceIndex
    n, d, nq, k = 150000, 512 if dtype == "f16" else 1024, 200, 100
    m, qs = corpus_and_query("gaussian", 4321, n, d, nq)
    rng = np.random.default_rng(9)
    for base, cnt, qi in ((70000, 300, 5), (70400, 150, 6), (20, 200, 150)):
        noise = rng.standard_normal((cnt, d)).astype(np.float32) * np.float32(0.15 / np.sqrt(d))
        rows = qs[qi][None, :] + noise
        m[base:base + cnt] = rows / np.linalg.norm(rows, axis=1, keepdims=True)
    idx = DeviceIndex(m, dtype=dtype)
    s0, r0 = idx.search_batch(qs, k)
    idx.set_variant(2)
    s2, r2 = idx.search_batch(qs, k)
    idx.set_variant(0)
    # nobody leaves the fused path: what does not fit a wave's eighth goes straight to the global lists
    assert np.array_equal(r0, r2) and np.array_equal(s0, s2)
    left = []
    md = idx.stored_rows()
    for qi in sorted(set(left) | {5, 6, 150, 0, 199}):
        qd = idx.stored_query(qs[qi])
        exp = oracle.cpu_search(md, qd, k)
        assert_topk_parity(s0[qi], r0[qi], [s for s, _ in exp], [i for _, i in exp],
                           oracle.cpu_scores_f64(md, qd), label=f"clustered {dtype} q{qi}")
    assert set(r0[5]) <= set(range(70000, 70300)) and set(r0[150]) <= set(range(20, 220))
    idx.release()


@pytest.mark.parametrize("dtype,k", [("f16", 100), ("f16", 8), ("fp8", 8)])
def test_phased_small_topics_flush_by_runs(gpu, dtype, k):
    """Topics of a few dozen rows: a wave parks FEW candidates in such a tile, all of one query -- the flush then adds per
    run of a query instead of per candidate (gemm_phased.h flush_group), switched on by one lane's survivor count (k = 100:
    the sweeps count per lane) or by what a wave on the compare-and-branch path parked (k = 8: it expected almost none).
    Several topics per tile, topics across tile and wave boundaries, two topics of the same query; the oracle's answer, and
    the round-1 kernel's bit for bit."""
    from svs_amd import DeviceIndex
    n, d, nq = 150000, 512 if dtype == "f16" else 1024, 200
    m, qs = corpus_and_query("gaussian", 777, n, d, nq)
    rng = np.random.default_rng(10)
    topics = ((90000, 40, 7), (90100, 24, 9), (90240, 40, 11), (120000 + 250, 12, 13), (256 * 300 + 120, 16, 7), (33, 60, 190))
    for base, cnt, qi in topics:
        noise = rng.standard_normal((cnt, d)).astype(np.float32) * np.float32(0.15 / np.sqrt(d))
        rows = qs[qi][None, :] + noise
        m[base:base + cnt] = rows / np.linalg.norm(rows, axis=1, keepdims=True)
    idx = DeviceIndex(m, dtype=dtype)
    s0, r0 = idx.search_batch(qs, k)
    idx.set_variant(2)
    s2, r2 = idx.search_batch(qs, k)
    idx.set_variant(0)
    assert np.array_equal(r0, r2) and np.array_equal(s0, s2)
    md = idx.stored_rows()
    for qi in (7, 9, 11, 13, 190, 0, 199):
        qd = idx.stored_query(qs[qi])
        exp = oracle.cpu_search(md, qd, k)
        assert_topk_parity(s0[qi], r0[qi], [s for s, _ in exp], [i for _, i in exp],
                           oracle.cpu_scores_f64(md, qd), label=f"small topics {dtype} k={k} q{qi}")
    own = set(range(90000, 90040)) | set(range(256 * 300 + 120, 256 * 300 + 136))
    assert set(r0[7][:min(k, 56)]) <= own
    idx.release()


def test_phased_kernel_rate_floor(gpu):
    """A tripwire, not a benchmark: the fused f16 panel kernel must run above 500 TFLOP/s (it measures
    ~1,250).  hipcc has twice turned a small source change into accumulators kept in scratch memory or
    hundreds of spilled registers -- correct results, 20x slower -- and parity tests cannot see that."""
    import torch
    from svs_amd import DeviceIndex
    n, d, nq = 262144, 1536, 1024
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    m = torch.randn((n, d), device=dev, generator=g)
    m /= m.norm(dim=1, keepdim=True)
    idx = DeviceIndex.from_device_pointer(m.data_ptr(), n, d, device=0, dtype="f16")
    del m
    q = torch.randn((nq, d), device=dev, generator=g)
    q = (q / q.norm(dim=1, keepdim=True)).cpu().numpy()
    for _ in range(3):
        idx.search_batch(q, 100)
    idx.set_timing(True)
    for _ in range(6):
        idx.search_batch(q, 100)
    _, _, cnt = idx.get_timing()
    ms = idx.last_dominant_ms_sum / cnt
    idx.release()
    tflops = 2.0 * n * d * nq / (ms * 1e-3) / 1e12
    print(f"phased f16 kernel: {ms:.3f} ms, {tflops:.0f} TFLOP/s")
    assert tflops > 500, f"{tflops:.0f} TFLOP/s: look at the kernel's register / scratch statistics"
