"""GPU: incremental update of the HBM-resident corpus (svs_index_append /
svs_index_mask_rows) -- results must equal those of an index
rebuilt from the edited matrix."""
import threading

import numpy as np
import pytest

from compare import assert_topk_parity
from oracle import svs_oracle as oracle
from synth import corpus_and_query

pytestmark = pytest.mark.gpu


def _same_as_rebuilt(idx, m_live, live_rows, qs, k):
    """idx (with tombstones) vs the oracle on the compacted matrix; rows map through live_rows."""
    for q in qs:
        got = idx.search(q, k)
        exp = oracle.cpu_search(m_live, q, k)
        assert len(got) == len(exp)
        assert_topk_parity([s for s, _ in got], [r for _, r in got],
                           [s for s, _ in exp], [int(live_rows[i]) for _, i in exp],
                           None if False else _truth(m_live, live_rows, q), label="update")


def _truth(m_live, live_rows, q):
    t = np.full(int(live_rows.max()) + 1, -np.inf)
    t[live_rows] = oracle.cpu_scores_f64(m_live, q)
    return t


@pytest.mark.parametrize("dtype,d", [("f32", 1536), ("f32", 100), ("f16", 1536), ("fp8", 1024)])
def test_append_then_mask(gpu, dtype, d):
    from svs_amd import DeviceIndex
    m, qs = corpus_and_query("gaussian", 90 + d, 30000, d, 3)
    idx = DeviceIndex(m[:8000], dtype=dtype)
    idx.append(m[8000:8001])                 # 1 row
    idx.append(m[8001:20000])                # forces the buffers to grow
    idx.append(m[20000:])
    assert idx.shape == (30000, d)
    ref = DeviceIndex(m, dtype=dtype)        # the same corpus uploaded in one go
    for q in qs:
        assert idx.search(q, 100) == ref.search(q, 100)       # bit-identical: same kernels, same rows
    bs, br = idx.search_batch(qs, 50)
    rs, rr = ref.search_batch(qs, 50)
    assert np.array_equal(bs, rs) and np.array_equal(br, rr)
    # tombstones
    rng = np.random.default_rng(4)
    dead = np.unique(np.concatenate([rng.choice(30000, 700, replace=False), [0, 29999],
                                     [r for _, r in ref.search(qs[0], 20)]]))
    idx.mask_rows(dead)
    idx.mask_rows(dead[:10])                 # masking twice is a no-op
    assert idx.n_masked == len(dead)
    live = np.setdiff1d(np.arange(30000), dead)
    md = ref.stored_rows()[live]             # what the index holds, compacted
    for q in qs:
        qd = ref.stored_query(q)
        got = idx.search(q, 100)
        exp = oracle.cpu_search(md, qd, 100)
        truth = np.full(30000, -np.inf); truth[live] = oracle.cpu_scores_f64(md, qd)
        assert_topk_parity([s for s, _ in got], [r for _, r in got], [s for s, _ in exp],
                           [int(live[i]) for _, i in exp], truth, label=f"{dtype} masked")
        assert not set(r for _, r in got) & set(dead.tolist())
    assert len(idx.search(qs[0], 10 ** 6)) == len(live)      # count = min(k, live rows)
    idx.release(); ref.release()


def test_mask_pairs_and_small_index(gpu):
    from svs_amd import DeviceIndex
    m, qs = corpus_and_query("gaussian", 5, 400, 64, 1)
    idx = DeviceIndex(m)
    idx.mask_rows([7, 100, 399])
    live = np.setdiff1d(np.arange(400), [7, 100, 399])
    got = idx.top_pairs(500)
    exp = oracle.cpu_top_pairs(np.dot(m[live], m[live].T), 500)
    assert [(int(live[i]), int(live[j])) for _, i, j in exp] == [(i, j) for _, i, j in got]
    s = idx.search(qs[0], 400)
    assert len(s) == 397 and not {7, 100, 399} & {r for _, r in s}
    with pytest.raises(ValueError):
        idx.mask_rows([400])
    with pytest.raises(ValueError):
        idx.append(np.zeros((2, 63), dtype=np.float32))
    idx.release()


def test_append_while_searching(gpu):
    """append() takes the geometry lock exclusively; concurrent searches see either the
    old or the new corpus, never a torn one."""
    from svs_amd import DeviceIndex
    m, qs = corpus_and_query("gaussian", 6, 60000, 256, 4)
    idx = DeviceIndex(m[:20000])
    before = [idx.search(q, 30) for q in qs]
    full = DeviceIndex(m)
    after = [full.search(q, 30) for q in qs]
    full.release()
    errs = []

    def searcher():
        try:
            for it in range(60):
                for qi, q in enumerate(qs):
                    r = idx.search(q, 30)
                    # (not `row < idx.n`: the Python attribute is refreshed AFTER svs_index_append returns, so a search
                    #  that runs in between rightly reports rows the attribute does not cover yet.  What must hold for
                    #  either corpus: a real row of the final matrix, with that row's own score, best first.)
                    rows = np.array([row for _, row in r])
                    sc = np.array([s for s, _ in r], dtype=np.float64)
                    assert len(r) == 30 and (rows >= 0).all() and (rows < len(m)).all() and len(set(rows.tolist())) == 30
                    assert np.abs(m[rows].astype(np.float64) @ q.astype(np.float64) - sc).max() < 1e-5
                    assert (np.diff(sc) <= 0).all()
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=searcher) for _ in range(4)]
    for t in ts:
        t.start()
    for c0 in range(20000, 60000, 5000):
        idx.append(m[c0:c0 + 5000])
    for t in ts:
        t.join()
    assert not errs, errs
    assert [idx.search(q, 30) for q in qs] == after and before != after
    idx.release()


@pytest.mark.parametrize("dtype,d,nq", [("f32", 1536, 16), ("f32", 512, 40), ("f16", 1536, 32), ("fp8", 1024, 64)])
def test_masked_rows_keep_the_fused_topk(gpu, dtype, d, nq):
    """One tombstone used to switch every batched search back to the materialised path (an
    nq x n score matrix per context).  Now the fused epilogue stays on: the prefix thresholds
    come from live rows only and select_final strikes masked candidates out.  >= 131,072 rows,
    >= 16 queries: equals the oracle on the compacted matrix and the materialised run of the
    same kernels (set_variant(6)) bit for bit -- and no nq x n matrix is allocated."""
    from svs_amd import DeviceIndex, _native
    n, k = 140000, 100
    m, qs = corpus_and_query("gaussian", 700 + d + nq, n, d, nq)
    idx = DeviceIndex(m, dtype=dtype)
    clean_s, clean_r = idx.search_batch(qs, k)
    rng = np.random.default_rng(5)
    dead = np.unique(np.concatenate([
        rng.choice(n, 3000, replace=False),
        clean_r[:, :30].ravel(),              # the 30 best rows of every query: winners must be replaced
        np.arange(0, 2000, 3),                # a good part of the prefix that seeds the thresholds
        [0, n - 1]]))
    idx.mask_rows(dead)
    free_before, _ = _native.device_memory(0)
    fs, fr = idx.search_batch(qs, k)
    free_after, _ = _native.device_memory(0)
    # the materialised path would have grown the context's score buffer to nq * n * 4 bytes
    assert free_before - free_after < nq * n * 4 // 2, "fused path fell back to a materialised score matrix"
    assert not np.isin(fr, dead).any()
    live = np.setdiff1d(np.arange(n), dead)
    md = idx.stored_rows()[live]
    for qi in range(0, nq, max(1, nq // 8)):
        qd = idx.stored_query(qs[qi])
        exp = oracle.cpu_search(md, qd, k)
        truth = np.full(n, -np.inf); truth[live] = oracle.cpu_scores_f64(md, qd)
        assert_topk_parity(fs[qi], fr[qi], [s for s, _ in exp], [int(live[i]) for _, i in exp], truth,
                           label=f"{dtype} masked fused q{qi}")
    idx.set_variant(6)
    ms, mr = idx.search_batch(qs, k)
    assert np.array_equal(mr, fr) and np.array_equal(ms, fs)
    idx.set_variant(0)
    # rows appended after masking are searchable and the old tombstones still hold
    extra = m[dead[:50]]
    idx.append(extra)
    s2, r2 = idx.search_batch(qs, k)
    assert not np.isin(r2, dead).any()
    ref_rows = np.concatenate([md, idx.stored_rows(n, 50)])
    ref_ids = np.concatenate([live, np.arange(n, n + 50)])
    for qi in (0, nq - 1):
        qd = idx.stored_query(qs[qi])
        exp = oracle.cpu_search(ref_rows, qd, k)
        truth = np.full(n + 50, -np.inf); truth[ref_ids] = oracle.cpu_scores_f64(ref_rows, qd)
        assert_topk_parity(s2[qi], r2[qi], [s for s, _ in exp], [int(ref_ids[i]) for _, i in exp], truth,
                           label=f"{dtype} masked fused + append q{qi}")
    idx.release()


@pytest.mark.parametrize("dtype", ["f32", "f16", "fp8"])
def test_build_from_device_blocks(gpu, dtype):
    """svs_index_create(NULL, 0, d) + svs_index_reserve + svs_index_append_from_device: an index
    assembled from device blocks equals the one uploaded from the host in one go, bit for bit."""
    import torch
    from svs_amd import DeviceIndex
    n, d = 50000, 1000            # d = 1000: rows padded to ld = 1024 (f32) -- the 2-D copy path
    m, qs = corpus_and_query("gaussian", 61, n, d, 20)
    ref = DeviceIndex(m, dtype=dtype)
    idx = DeviceIndex.empty(d, dtype=dtype, reserve=30000)
    assert idx.shape == (0, d)
    with pytest.raises(ValueError):
        idx.search(qs[0], 5)       # empty matrix: the reference raises ValueError too
    for r0, r1 in ((0, 1), (1, 20000), (20000, 30000), (30000, 50000)):   # the last block outgrows the reservation
        blk = torch.from_numpy(m[r0:r1]).to("cuda:0")
        idx.append_device(blk.data_ptr(), r1 - r0)
        del blk
    assert idx.shape == (n, d) and idx.hbm_bytes >= ref.hbm_bytes
    assert np.array_equal(idx.stored_rows(), ref.stored_rows())
    for q in qs[:3]:
        assert idx.search(q, 100) == ref.search(q, 100)
    a, b = idx.search_batch(qs, 50), ref.search_batch(qs, 50)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # strided device source
    wide = torch.zeros((100, d + 24), device="cuda:0")
    wide[:, :d] = torch.from_numpy(m[:100]).to("cuda:0")
    idx.append_device(wide.data_ptr(), 100, src_ld=d + 24)
    assert np.array_equal(idx.stored_rows(n, 100), ref.stored_rows(0, 100))
    with pytest.raises(ValueError):
        idx.append_device(wide.data_ptr(), 10, src_ld=d - 1)
    idx.release(); ref.release()


def test_scores_after_append_through_another_owner(gpu):
    """`scores()` on a wrapper whose row count is stale (the rows were appended through another owner of the same
    handle): svs_index_scores_n refuses the short buffer, reports the handle's row count, and the wrapper repeats the
    call with it -- round 3's capacity-less entry wrote past the buffer here (gpurun_out/r3c_tests.log)."""
    from svs_amd import DeviceIndex
    m, qs = corpus_and_query("gaussian", 321, 9000, 200, 1)
    idx = DeviceIndex(m[:5000])
    other = idx.share()
    other.append(m[5000:])
    assert idx.n == 5000 and other.n == 9000          # this wrapper has not looked since
    got = idx.scores(qs[0])
    assert got.shape == (9000,)
    np.testing.assert_allclose(got, oracle.cpu_scores(m, qs[0]), atol=1e-5, rtol=0)
    # the C entry itself: a short capacity is an error that names the row count, nothing is written
    import ctypes as C
    from svs_amd import _native
    lib = _native.load()
    buf = np.full(5000 + 16, 7.0, dtype=np.float32)
    now = C.c_int64(0)
    rc = lib.svs_index_scores_n(idx._handle(), qs[0].ctypes.data_as(C.c_void_p), 200, buf.ctypes.data_as(C.c_void_p), 5000, C.byref(now))
    assert rc == _native.SVS_ERR_INVALID and now.value == 9000 and (buf == 7.0).all()
    other.release()
    idx.release()


def test_append_while_fused_batches_run(gpu):
    """Rows are appended (the fused path's threshold sample is re-taken after every append: svs_amd.hip prefix_image)
    while three threads keep 16-query batches on the fused path in flight: every answer must be a real, distinct set of
    rows of the final matrix with those rows' own scores, best first -- for whichever row count the search saw -- and
    once the appends are done the answers equal a fresh index over everything."""
    from svs_amd import DeviceIndex
    rng = np.random.default_rng(33)
    n0, n1, d, nq, k = 140000, 200000, 128, 16, 40
    m = rng.standard_normal((n1, d)).astype(np.float32)
    m /= np.linalg.norm(m, axis=1, keepdims=True)
    qs = rng.standard_normal((nq, d)).astype(np.float32)
    qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    idx = DeviceIndex.empty(d, reserve=n0)          # (no room reserved for the appends: the buffers are re-allocated too)
    idx.append(m[:n0])
    errs, stop = [], threading.Event()

    def searcher():
        try:
            while not stop.is_set():
                bs, br = idx.search_batch(qs, k)
                assert br.shape == (nq, k) and (br >= 0).all() and (br < n1).all()
                for qi in (0, nq - 1):
                    rows = br[qi]
                    assert len(set(rows.tolist())) == k
                    assert np.abs(m[rows].astype(np.float64) @ qs[qi].astype(np.float64) - bs[qi]).max() < 1e-5
                    assert (np.diff(bs[qi].astype(np.float64)) <= 0).all()
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=searcher) for _ in range(3)]
    for t in ts:
        t.start()
    for c0 in range(n0, n1, 6000):
        idx.append(m[c0:c0 + 6000])
    stop.set()
    for t in ts:
        t.join()
    assert not errs, errs
    full = DeviceIndex(m)
    bs, br = idx.search_batch(qs, k)
    fs, fr = full.search_batch(qs, k)
    assert np.array_equal(br, fr) and np.array_equal(bs, fs)
    full.release()
    idx.release()
