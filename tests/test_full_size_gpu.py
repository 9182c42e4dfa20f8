"""GPU: size-independent properties at BASELINE.json's full sizes, where the numpy
oracle would take minutes per query (the oracle-checked cases at these sizes are the
golden 1M x 1536 single-query vectors in test_search_gpu.py).

  * batch == loop: a batched search is by definition the per-query searches in order
    (the reference loops np.dot, src/svs/kb.py:1623);
  * fused == materialised: the fused top-k epilogue (no score matrix) must return exactly
    what the same kernel returns when every score is written and selected from
    (svs_index_set_variant(6) turns the fusion off) -- bit for bit, same summation order;
  * a planted row is found: a corpus row equal to the query scores 1 and comes first.

Corpora are generated on the device (torch is plumbing here: device memory and RNG)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _corpus(n, d, seed, block=250_000):
    import torch
    dev = torch.device("cuda:0")
    out = torch.empty((n, d), device=dev, dtype=torch.float32)
    for b0 in range(0, n, block):
        g = torch.Generator(device=dev)
        g.manual_seed(seed * 1_000_003 + b0)
        x = torch.randn((min(block, n - b0), d), device=dev, dtype=torch.float32, generator=g)
        x /= x.norm(dim=1, keepdim=True)
        out[b0:b0 + x.shape[0]] = x
        del x
    return out


def _queries(nq, d, seed):
    import torch
    g = torch.Generator(device="cuda:0")
    g.manual_seed(seed)
    q = torch.randn((nq, d), device="cuda:0", dtype=torch.float32, generator=g)
    q /= q.norm(dim=1, keepdim=True)
    return q


def _index(rows, dtype):
    import torch
    from svs_amd import DeviceIndex
    idx = DeviceIndex.from_device_pointer(rows.data_ptr(), rows.shape[0], rows.shape[1], device=0, dtype=dtype)
    torch.cuda.synchronize()
    return idx


def _same_up_to_near_ties(s_a, r_a, s_b, r_b, tol):
    """Two exact top-k lists computed in different summation orders: scores agree within
    tol, and rows agree wherever the neighbouring scores are further apart than tol."""
    assert np.max(np.abs(s_a - s_b)) <= tol
    diff = np.nonzero(r_a != r_b)[0]
    for i in diff:
        lo, hi = max(0, i - 1), min(len(s_a) - 1, i + 1)
        assert min(abs(s_a[i] - s_a[lo]) if lo != i else 1.0, abs(s_a[i] - s_a[hi]) if hi != i else 1.0) <= 2 * tol, \
            f"rank {i}: rows {r_a[i]} vs {r_b[i]} differ without a near tie"


def test_config1_f32_batch16_equals_single_queries(gpu):
    """BASELINE configs[1] size (1M x 1536 f32): 16 queries per corpus pass (4x4x1 MFMA kernel,
    fused top-k) against 16 single-query searches (GEMV + materialised top-k)."""
    import torch
    n, d, k = 1_000_000, 1536, 100
    rows = _corpus(n, d, 4242)
    qs = _queries(16, d, 99)
    planted = 777_777
    qs[3] = rows[planted]
    idx = _index(rows, "f32")
    del rows
    torch.cuda.empty_cache()
    qh = qs.cpu().numpy()
    bs, br = idx.search_batch(qh, k)
    assert bs.shape == (16, k)
    for qi in range(16):
        one = idx.search(qh[qi], k)
        s1 = np.array([s for s, _ in one], dtype=np.float32)
        r1 = np.array([r for _, r in one], dtype=np.int64)
        _same_up_to_near_ties(bs[qi], br[qi], s1, r1, 2e-6)
        assert np.all(np.diff(bs[qi]) <= 0)
    assert br[3, 0] == planted and abs(bs[3, 0] - 1.0) < 1e-5
    # and the fused epilogue loses nothing: materialised run of the same kernel, bit for bit
    idx.set_variant(6)
    ms, mr = idx.search_batch(qh, k)
    assert np.array_equal(mr, br) and np.array_equal(ms, bs)
    idx.release()


def test_config2_f16_b1024_fused_equals_materialised(gpu):
    """BASELINE configs[2]: 1M x 1536 f16, 1024 queries per call.  The fused run never writes
    the 4 GB score matrix; the materialised run of the same GEMM does, 256 queries at a time."""
    import torch
    n, d, k, nq = 1_000_000, 1536, 100, 1024
    rows = _corpus(n, d, 5151)
    qs = _queries(nq, d, 17)
    planted = 123_456
    qs[1000] = rows[planted]
    idx = _index(rows, "f16")
    del rows
    torch.cuda.empty_cache()
    qh = qs.cpu().numpy()
    fs, fr = idx.search_batch(qh, k)
    assert fr[1000, 0] == planted and abs(fs[1000, 0] - 1.0) < 2e-3      # f16 rounding of a unit vector
    assert np.all(np.diff(fs, axis=1) <= 0) and fr.min() >= 0 and fr.max() < n
    idx.set_variant(6)
    for q0 in range(0, nq, 256):
        ms, mr = idx.search_batch(qh[q0:q0 + 256], k)
        assert np.array_equal(mr, fr[q0:q0 + 256]) and np.array_equal(ms, fs[q0:q0 + 256]), f"queries {q0}.."
    # a single-query search (GEMV kernel, other summation order) agrees up to near ties
    idx.set_variant(0)
    for qi in (0, 511, 1000, 1023):
        one = idx.search(qh[qi], k)
        _same_up_to_near_ties(fs[qi], fr[qi], np.array([s for s, _ in one], dtype=np.float32),
                              np.array([r for _, r in one], dtype=np.int64), 4e-6)
    idx.release()


def test_config4_fp8_b256_fused_equals_materialised(gpu):
    """BASELINE configs[4] shape (dim 3072, fp8 corpus, 256 queries per call) at 2.5M rows --
    a quarter of its 10M, which is generated in 123 GB of f32 first and takes minutes; the
    prefix threshold (n / 64 rows) and the 256 x 256 tiles are exercised the same way."""
    import torch
    n, d, k, nq = 2_500_000, 3072, 100, 256
    rows = _corpus(n, d, 6262)
    qs = _queries(nq, d, 23)
    planted = 2_400_001
    qs[200] = rows[planted]
    idx = _index(rows, "fp8")
    del rows
    torch.cuda.empty_cache()
    qh = qs.cpu().numpy()
    fs, fr = idx.search_batch(qh, k)
    assert fr[200, 0] == planted and abs(fs[200, 0] - 1.0) < 0.02        # e4m3 rounding
    assert np.all(np.diff(fs, axis=1) <= 0)
    idx.set_variant(6)
    ms, mr = idx.search_batch(qh, k)
    assert np.array_equal(mr, fr) and np.array_equal(ms, fs)
    idx.release()
