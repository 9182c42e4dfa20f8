"""GPU: fp8-resident corpus (SVS_DTYPE_FP8, OCP e4m3fn + per-row f32 scale;
BASELINE.json configs[4]).  Oracle = numpy's f32 path on the DEQUANTISED corpus and
query, read back from the index (what is really stored), SURVEY.md 7 hard part 7."""
import numpy as np
import pytest

from compare import assert_topk_parity
from oracle import svs_oracle as oracle
from synth import corpus_and_query

pytestmark = pytest.mark.gpu


def test_fp8_quantisation_matches_ocp_e4m3(gpu):
    """Stored bytes == RNE rounding to e4m3fn of row / (max|row| / 448), checked with
    torch's CPU float8_e4m3fn."""
    import torch
    from svs_amd import DeviceIndex
    m, _ = corpus_and_query("gaussian", 3, 500, 200, 1)
    idx = DeviceIndex(m, dtype="fp8")
    got = idx.stored_rows()
    scale = (np.abs(m).max(axis=1, keepdims=True) / np.float32(448.0)).astype(np.float32)
    q = torch.from_numpy((m * (np.float32(1.0) / scale)).astype(np.float32)).to(torch.float8_e4m3fn).to(torch.float32).numpy()
    exp = q * scale
    mism = np.mean(got != exp)
    assert mism < 1e-3, f"{mism:.2%} of elements differ from the e4m3fn reference"   # ties of 1/scale rounding only
    assert np.max(np.abs(got - m)) < 0.07 * np.abs(m).max()                           # 3 mantissa bits
    idx.release()


def _e4m3_external(x):
    """x (.., d) f32 -> (dequantised f32, scale): the library's recipe restated with torch's CPU
    float8_e4m3fn cast (RNE): scale = max|x| / 448 per vector, q = e4m3(x * (1 / scale))."""
    import torch
    x = np.asarray(x, dtype=np.float32)
    scale = (np.abs(x).max(axis=-1, keepdims=True) / np.float32(448.0)).astype(np.float32)
    q = torch.from_numpy((x * (np.float32(1.0) / scale)).astype(np.float32)).to(torch.float8_e4m3fn).to(torch.float32).numpy()
    return q * scale, scale


def test_fp8_query_quantiser_external(gpu):
    """The QUERY side of the fp8 path against something that is not the library: torch's CPU
    e4m3fn cast.  Vectors are built so that their scale is an exact power of two (max|x| =
    448 * 2^-12): 1 / scale and every x / scale are then exact in f32, so there are no rounding
    ties of the division left and the comparison is bit for bit.  Checked three ways: the
    quantised query read back (stored_query), the scores the single-query kernel returns, and the
    top-k of the batched MFMA kernel -- both against numpy on the EXTERNALLY quantised corpus and
    query, so a quantiser bug on either side shows as a score difference of ~1e-4, not 1e-5."""
    from svs_amd import DeviceIndex
    top = np.float32(448.0 * 2.0 ** -12)
    m, qs = corpus_and_query("gaussian", 77, 6000, 512, 20)
    m = np.clip(m, -top, top)
    qs = np.clip(qs, -top, top)
    m[:, 3] = top
    qs[:, 5] = -top
    idx = DeviceIndex(m, dtype="fp8")
    rows_ext, rscale = _e4m3_external(m)
    assert np.all(rscale == np.float32(2.0 ** -12))
    assert np.array_equal(idx.stored_rows(), rows_ext)
    q_ext = []
    for q in qs:
        qe, qscale = _e4m3_external(q)
        assert qscale[0] == np.float32(2.0 ** -12)
        assert np.array_equal(idx.stored_query(q), qe)          # the query quantiser, bit for bit
        q_ext.append(qe)
        got = idx.scores(q)                                      # what the kernel really multiplies
        exp = rows_ext.astype(np.float64) @ qe.astype(np.float64)
        assert np.max(np.abs(got - exp)) <= 1e-5, np.max(np.abs(got - exp))
    bs, br = idx.search_batch(qs, 50)                            # MFMA path, 20 queries
    for qi, qe in enumerate(q_ext):
        exp = oracle.cpu_search(rows_ext, qe, 50)
        assert_topk_parity(bs[qi], br[qi], [s for s, _ in exp], [i for _, i in exp],
                           oracle.cpu_scores_f64(rows_ext, qe), label=f"fp8 external q{qi}")
    idx.release()


@pytest.mark.parametrize("n,d,k", [(20000, 3072, 100), (20000, 1536, 100), (6000, 1024, 10), (5000, 100, 7),
                                   (600, 3, 5), (50, 1537, 100)])
def test_fp8_single_query(gpu, n, d, k):
    from svs_amd import DeviceIndex
    m, qs = corpus_and_query("gaussian", 300 + n, n, d, 3)
    idx = DeviceIndex(m, dtype="fp8")
    assert idx.dtype == "fp8" and idx.ld % 16 == 0 and idx.ld >= d and idx.hbm_bytes == n * idx.ld + 4 * n
    md = idx.stored_rows()
    for q in qs:
        qd = idx.stored_query(q)
        got = idx.search(q, k)
        exp = oracle.cpu_search(md, qd, k)
        assert_topk_parity([s for s, _ in got], [i for _, i in got], [s for s, _ in exp], [i for _, i in exp],
                           oracle.cpu_scores_f64(md, qd), label=f"fp8 {n}x{d}")
    idx.release()


@pytest.mark.parametrize("n,d,nq,k", [(20000, 3072, 32, 100), (20000, 1536, 33, 100), (9000, 3072, 256, 50),
                                      (5000, 128, 7, 100), (140000, 768, 64, 100), (135000, 1536, 17, 100),
                                      (140000, 3072, 40, 50)])
def test_fp8_batch_mfma(gpu, n, d, nq, k):
    """Batched path: e4m3 operands on v_mfma_f32_16x16x128_f8f6f4 (the cases with >= 16 queries
    over >= 131,072 rows also take the fused top-k epilogue)."""
    from svs_amd import DeviceIndex
    m, qs = corpus_and_query("gaussian", 600 + n + nq, n, d, nq)
    idx = DeviceIndex(m, dtype="fp8")
    md = idx.stored_rows()
    bs, br = idx.search_batch(qs, k)
    for qi in range(0, nq, max(1, nq // 12)):
        qd = idx.stored_query(qs[qi])
        exp = oracle.cpu_search(md, qd, k)
        assert_topk_parity(bs[qi], br[qi], [s for s, _ in exp], [i for _, i in exp],
                           oracle.cpu_scores_f64(md, qd), label=f"fp8 batch {n}x{d} q{qi}")
    idx.release()


def test_fp8_recall_vs_f32(gpu):
    """recall@100 of the fp8 index against the f32 truth (reported, loosely bounded)."""
    from svs_amd import DeviceIndex
    m, qs = corpus_and_query("gaussian", 4243, 50000, 3072, 8)
    idx = DeviceIndex(m, dtype="fp8")
    rec = []
    for q in qs:
        got = {i for _, i in idx.search(q, 100)}
        exp = {i for _, i in oracle.cpu_search(m, q, 100)}
        rec.append(len(got & exp) / 100)
    idx.release()
    print("fp8 recall@100 vs f32:", rec)
    assert np.mean(rec) >= 0.80, rec
