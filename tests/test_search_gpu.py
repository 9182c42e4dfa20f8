"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors
recorded from the reference and against the numpy oracle on the same inputs."""
import json
import os
import threading

import numpy as np
import pytest

from compare import assert_topk_parity
from oracle import svs_oracle as oracle
from synth import corpus_and_query

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def _index(m):
    from svs_amd import DeviceIndex
    return DeviceIndex(m)


# ---- A4 through the device: a (N, 1) corpus with q = [1.0] makes score == value
def test_get_top_k_golden_cases_through_hip(gpu):
    g = _load("topk_cases.json")
    ran = 0
    for c in g["cases"]:
        if c["dtype"] != "float32" or len(c["scores"]) == 0:
            continue
        arr = np.array(c["scores"], dtype=np.float32)
        idx = _index(arr[:, None])
        got = idx.search(np.array([1.0], dtype=np.float32), c["k"])
        assert [[s, i] for s, i in got] == c["expected"], c["note"]
        idx.release()
        ran += 1
    assert ran > 30


def test_get_top_k_boundary_ties(gpu):
    for c in _load("topk_cases.json")["boundary_ties"]:
        arr = np.array(c["scores"], dtype=np.float32)
        idx = _index(arr[:, None])
        got = idx.search(np.array([1.0], dtype=np.float32), c["k"])
        assert [s for s, _ in got] == c["expected_scores"]
        # our documented rule at a boundary tie: the largest rows win
        assert got == oracle.total_order_top_k(arr, c["k"])
        idx.release()


@pytest.mark.parametrize("case", _load("search_cases.json")["cases"],
                         ids=lambda c: f'{c["kind"]}-{c["n"]}x{c["d"]}-k{c["k"]}')
def test_search_golden(gpu, case):
    """Golden vectors recorded from the real reference.  Gate (north_star: "returned indices match
    the numpy reference bit-exact"): on every GAUSSIAN case -- BASELINE configs[0] and configs[1]
    among them -- the single-query path must reproduce the reference's row order position by
    position, swaps == 0, whatever the recorded gap.  (make_golden's check: on all of these the
    reference's order equals the order of the correctly rounded f64 scores, so nothing about them
    is a coin flip of numpy's.)  The `uniform` cases (the reference notebook's recipe: all scores
    inside [0.71, 0.78], adjacent gaps of ~3e-7 against numpy's own ~2e-7 error) were allowed explained
    swaps (compare.py) through round 3 and are held to 0 as well since round 4.  The batch entry and the coalesced route
    (MFMA summation order) are gated the same way: 0 differing positions."""
    import conftest
    m, qs = corpus_and_query(case["kind"], case["seed"], case["n"], case["d"], case["nq"])
    idx = _index(m)
    assert idx.shape == (case["n"], case["d"])
    truths = [oracle.cpu_scores_f64(m, q) for q in qs]
    single = 0
    for qi, q in enumerate(qs):
        got = idx.search(q, case["k"])
        single += assert_topk_parity([s for s, _ in got], [i for _, i in got],
                                     case["scores"][qi], case["rows"][qi], truths[qi],
                                     label=f'{case["note"]} q{qi}')
        if case["kind"] == "gaussian":
            assert [i for _, i in got] == case["rows"][qi], f'{case["note"]} q{qi}: row order differs from the reference'
    # batch entry (up to 16 queries share one corpus pass; MFMA summation order)
    batch = 0
    bs, br = idx.search_batch(qs, case["k"])
    for qi, q in enumerate(qs):
        batch += assert_topk_parity(bs[qi], br[qi], case["scores"][qi], case["rows"][qi], truths[qi],
                                    label=f'{case["note"]} batch q{qi}')
    # the route callers get under load (svs_index_set_coalesce, on by default behind KB.retrieve: the executor
    # threads of reference src/svs/kb.py:1184-1190): passes of chosen sizes, formed out of concurrent single-query
    # calls, answered by the 16-query streaming kernel (2 .. 16), the tiled MFMA kernels (17 ..) and, from
    # 131,072 rows and 16 queries up, their fused top-k epilogues.  Every caller's answer against the fixture.
    coalesced = 0
    idx.set_coalesce(True)
    before = idx.coalesce_sizes()
    sizes = (2, 7, 16, 32, 64) if case["n"] > 1 else (2,)
    for size in sizes:
        got = [None] * size
        idx.coalesce_hold(size)

        def caller(t):
            got[t] = idx.search(qs[t % len(qs)], case["k"])

        ts = [threading.Thread(target=caller, args=(t,)) for t in range(size)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        for t in range(size):
            qi = t % len(qs)
            coalesced += assert_topk_parity([s for s, _ in got[t]], [i for _, i in got[t]], case["scores"][qi], case["rows"][qi],
                                            truths[qi], label=f'{case["note"]} coalesced pass of {size}, caller {t}')
    after = idx.coalesce_sizes()
    made = {s: after.get(s, 0) - before.get(s, 0) for s in after}
    for size in sizes:
        assert made.get(size, 0) >= 1, f"no pass of {size} queries was formed: {made}"
    idx.set_coalesce(False)
    idx.release()
    gaps = [g for g in case["min_adjacent_gap_f64"] if g is not None]
    conftest.PARITY_SWAPS[f'{case["kind"]} {case["n"]}x{case["d"]} k={case["k"]} ({case["note"]})'] = {
        "single": single, "batch": batch, "coalesced": coalesced, "queries": case["nq"], "min_gap": min(gaps) if gaps else None}
    # bit-exact row order on every route -- alone, as a batch, and coalesced with other callers -- on EVERY golden case:
    # since round 4 also on the `uniform` near-tie ones (adjacent f64 gaps of ~3.6e-7), which rounds 1-3 allowed to differ
    # by explained swaps and which never did (profiles/r*_parity_swaps.json: 0 on every route, every round)
    assert single == 0 and batch == 0 and coalesced == 0, (single, batch, coalesced)


def test_nan_scores_rank_largest(gpu):
    """A4: np.argpartition treats NaN as the largest score (reference src/svs/util.py:202), so a
    row whose score is NaN is always inside the top-k set.  The reference's ORDER around a NaN is
    whatever Python's sort makes of incomparable tuples (util.py:203); ours is the documented total
    order (NaN first, ties row desc).  Checked on every select path: direct (n <= 4096), windowed
    select, full sort (k > 2048), and both fused batch epilogues."""
    q = np.array([1.0], dtype=np.float32)
    rng = np.random.default_rng(21)

    def same(got, exp):
        assert [r for _, r in got] == [r for _, r in exp]
        gs, es = np.array([s for s, _ in got]), np.array([s for s, _ in exp])
        assert np.array_equal(np.isnan(gs), np.isnan(es)) and np.array_equal(gs[~np.isnan(gs)], es[~np.isnan(es)])

    for n, nan_rows in ((6, [1, 4]), (3000, [7]), (70000, [5, 69999, 31337]), (200000, list(range(1000, 1150)))):
        v = rng.standard_normal(n).astype(np.float32) * 0.3
        v[nan_rows] = np.nan
        idx = _index(v[:, None])
        for k in (1, 3, 100, 2048, 3000):
            got = idx.search(q, k)
            same(got, oracle.total_order_top_k(v, k))
            # the reference's set (defined when its k-th and (k+1)-th scores differ: not inside the NaNs)
            if len(nan_rows) <= k:
                assert {r for _, r in got} == {r for _, r in oracle.cpu_top_k(v, min(k, n))}
        idx.release()
    # batched kernels (fused epilogues from 16 queries and 131,072 rows up): a NaN inside one corpus row
    m, qs = corpus_and_query("gaussian", 31, 140000, 256, 32)
    m[[17, 70001, 139999], 5] = np.nan
    for dtype, nq in (("f32", 16), ("f32", 32), ("f16", 16), ("f16", 32)):
        from svs_amd import DeviceIndex
        idx = DeviceIndex(m, dtype=dtype)
        bs, br = idx.search_batch(qs[:nq], 10)
        for qi in range(nq):
            one = idx.search(qs[qi], 10)
            assert [r for _, r in one][:3] == [139999, 70001, 17] and list(br[qi][:3]) == [139999, 70001, 17], (dtype, nq, qi)
            assert np.all(np.isnan(bs[qi][:3])) and not np.any(np.isnan(bs[qi][3:]))
        idx.release()


def test_huge_k_is_clamped_not_truncated(gpu):
    """k is an int32 in the C ABI: 2**32 must not wrap to 0 and 2**31 must not go negative
    (reference src/svs/util.py:198-199 clamps top_k to len(scores) first)."""
    m, qs = corpus_and_query("gaussian", 41, 500, 32, 2)
    idx = _index(m)
    full = idx.search(qs[0], 500)
    for k in (2 ** 31 - 1, 2 ** 31, 2 ** 32, 2 ** 40):
        assert idx.search(qs[0], k) == full
        s, r = idx.search_batch(qs, k)
        assert s.shape == (2, 500) and list(r[0]) == [i for _, i in full]
    assert len(idx.top_pairs(2 ** 32)) == 500 * 499 // 2
    idx.release()


def test_scores_vector_matches_numpy(gpu):
    m, qs = corpus_and_query("gaussian", 5, 3000, 1536, 1)
    idx = _index(m)
    got = idx.scores(qs[0])
    exp = oracle.cpu_scores(m, qs[0])
    assert got.dtype == np.float32 and got.shape == exp.shape
    assert np.max(np.abs(got.astype(np.float64) - exp)) <= 1e-5
    truth = oracle.cpu_scores_f64(m, qs[0])
    # our f32 summation must be at least as close to f64 truth as 5e-7
    assert np.max(np.abs(got - truth)) < 5e-7
    idx.release()


@pytest.mark.parametrize("n,d", [(1, 1), (5, 2), (63, 5), (64, 7), (65, 12), (257, 33), (1000, 255),
                                 (1000, 257), (513, 512), (300, 1024), (100, 2048), (50, 4096), (40, 1280)])
def test_odd_shapes_against_oracle(gpu, n, d):
    m, qs = corpus_and_query("gaussian", 100 + n + d, n, d, 2)
    idx = _index(m)
    for q in qs:
        for k in (1, 7, n, n + 3):
            got = idx.search(q, k)
            exp = oracle.cpu_search(m, q, k)
            assert_topk_parity([s for s, _ in got], [i for _, i in got],
                               [s for s, _ in exp], [i for _, i in exp],
                               oracle.cpu_scores_f64(m, q), label=f"{n}x{d} k={k}")
    idx.release()


def test_full_ranking_path(gpu):
    """k > 1024 goes through the global bitonic sort; the reference's 'rank the
    whole KB' call (n = 10,548, examples/dad_jokes)."""
    m, qs = corpus_and_query("gaussian", 77, 10548, 256, 1)
    idx = _index(m)
    for k in (1025, 5000, 10548, 20000):
        got = idx.search(qs[0], k)
        exp = oracle.cpu_search(m, qs[0], k)
        assert len(got) == min(k, 10548)
        assert_topk_parity([s for s, _ in got], [i for _, i in got],
                           [s for s, _ in exp], [i for _, i in exp],
                           oracle.cpu_scores_f64(m, qs[0]), label=f"full ranking k={k}")
    idx.release()


def test_select_paths_agree(gpu):
    """path A (radix select, k <= 1024) and path B (full sort) must produce the
    same prefix on the same scores."""
    rng = np.random.default_rng(3)
    vals = rng.standard_normal(70000).astype(np.float32)
    idx = _index(vals[:, None])
    q = np.array([1.0], dtype=np.float32)
    a = idx.search(q, 1024)
    b = idx.search(q, 1500)
    assert a == b[:1024]
    assert a == oracle.total_order_top_k(vals, 1024)
    idx.release()


def test_mass_ties(gpu):
    q = np.array([1.0], dtype=np.float32)
    # every score identical: winners are the largest rows (documented rule)
    eq = np.full(20000, 0.25, dtype=np.float32)
    idx = _index(eq[:, None])
    assert idx.search(q, 5) == [(0.25, r) for r in (19999, 19998, 19997, 19996, 19995)]
    assert idx.search(q, 1000) == oracle.total_order_top_k(eq, 1000)
    idx.release()
    # 6000 ties straddling the k-th place (more than the candidate buffer holds)
    rng = np.random.default_rng(4)
    v = rng.standard_normal(50000).astype(np.float32) * 0.1
    tie_rows = rng.choice(50000, 6000, replace=False)
    v[tie_rows] = 0.9
    v[rng.choice(np.setdiff1d(np.arange(50000), tie_rows), 40, replace=False)] = 1.5
    idx = _index(v[:, None])
    for k in (10, 41, 100, 1024):
        assert idx.search(q, k) == oracle.total_order_top_k(v, k), k
    idx.release()
    # negative zero and positive zero tie (python: -0.0 == 0.0), order by row
    z = np.array([0.0, -0.0, 0.0, -0.0, -1.0], dtype=np.float32)
    idx = _index(z[:, None])
    got = idx.search(q, 4)
    assert [i for _, i in got] == [3, 2, 1, 0] and all(s == 0.0 for s, _ in got)
    idx.release()


def test_argument_semantics(gpu):
    from svs_amd import DeviceIndex
    m, qs = corpus_and_query("gaussian", 9, 100, 8, 1)
    idx = DeviceIndex(m)
    assert idx.search(qs[0], 0) == []
    assert idx.search(qs[0], -3) == []
    assert len(idx.search(qs[0], 1000)) == 100
    with pytest.raises(ValueError):       # numpy: shapes (100,8) and (7,) not aligned
        idx.search(qs[0][:7], 3)
    with pytest.raises(AssertionError):   # reference asserts isinstance(top_k, int)
        idx.search(qs[0], np.int64(3))
    idx.release()
    with pytest.raises(RuntimeError):
        idx.search(qs[0], 3)
    # empty table -> (0, 0) matrix -> the reference's retrieve() raises ValueError
    empty = DeviceIndex(np.zeros((0, 0), dtype=np.float32))
    with pytest.raises(ValueError):
        empty.search(np.array([1.0, 0.0, 0.0], dtype=np.float32), 1)
    empty.release()


def test_row_offset(gpu):
    from svs_amd import DeviceIndex
    m, qs = corpus_and_query("gaussian", 10, 5000, 64, 1)
    a = DeviceIndex(m)
    b = DeviceIndex(m, row_offset=1_000_000_000_000)
    ra, rb = a.search(qs[0], 10), b.search(qs[0], 10)
    assert [(s, i + 1_000_000_000_000) for s, i in ra] == rb
    a.release(); b.release()


def test_concurrent_searches_and_release(gpu):
    """AsyncKB runs superheavy() on executor threads outside its lock
    (reference src/svs/kb.py:1184-1190) and invalidate() can race with them."""
    m, qs = corpus_and_query("gaussian", 11, 40000, 256, 8)
    idx = _index(m)
    expected = [idx.search(q, 50) for q in qs]
    errors = []

    def worker(t):
        try:
            for it in range(20):
                qi = (t + it) % len(qs)
                assert idx.search(qs[qi], 50) == expected[qi]
        except RuntimeError as e:   # released underneath us: allowed, but only this error
            assert "released" in str(e)
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in ts:
        t.start()
    ts[0].join()
    idx.release()   # in-flight searches keep the HBM alive
    for t in ts:
        t.join()
    assert not errors, errors


def test_select_window_fallbacks(gpu):
    """The fast top-k path counts scores in a log-binned window [2^-31, 2]; outside
    it (negative / zero / huge scores, crowded bins) an exact in-kernel fallback
    takes over.  Every branch must agree with the rule-based oracle."""
    q = np.array([1.0], dtype=np.float32)
    rng = np.random.default_rng(8)
    cases = {
        "all negative": -np.abs(rng.standard_normal(30000).astype(np.float32)) - 0.01,
        "only 40 positive": np.concatenate([-np.abs(rng.standard_normal(29960)), np.abs(rng.standard_normal(40))]).astype(np.float32),
        "zeros and negatives": np.concatenate([np.zeros(5000), -np.abs(rng.standard_normal(20000))]).astype(np.float32),
        "scores above 2 (unnormalised)": (rng.standard_normal(30000) * 50).astype(np.float32),
        "crowded octave": (0.75 + rng.random(200000) * 1e-3).astype(np.float32),
        "tiny positives": (rng.random(30000) * 1e-12).astype(np.float32),
        "infinities": np.concatenate([rng.standard_normal(9000), [np.inf, -np.inf, np.inf]]).astype(np.float32),
    }
    for name, v in cases.items():
        v = rng.permutation(v)
        idx = _index(v[:, None])
        for k in (1, 100, 1000, 2048):
            assert idx.search(q, k) == oracle.total_order_top_k(v, k), (name, k)
        idx.release()


def test_no_hbm_leak_over_index_lifetimes(gpu):
    """create / search / release in a loop must give the HBM back (contexts, scratch,
    staging buffers and the corpus are all owned by the handle)."""
    from svs_amd import DeviceIndex, _native
    m, qs = corpus_and_query("gaussian", 1, 40000, 512, 20)
    free0 = None
    for it in range(12):
        idx = DeviceIndex(m, dtype=("f32", "f16", "fp8")[it % 3])
        idx.search(qs[0], 10)
        idx.search_batch(qs, 10)
        idx.top_pairs(5) if it == 0 else None
        idx.release()
        free, _ = _native.device_memory(0)
        if it == 2:
            free0 = free
    assert free0 - free < 64 << 20, f"HBM shrank by {(free0 - free) >> 20} MiB over 9 index lifetimes"
