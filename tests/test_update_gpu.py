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
                    mid_ok = all(row < idx.n for _, row in r)
                    assert mid_ok
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
