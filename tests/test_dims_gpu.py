"""GPU: the score stage at many row lengths.  A row is served by an exact-geometry kernel when it
is a whole number of 1 KiB wave loads, by gemv_unrolled.h (T lanes per row, NC chunks per lane)
otherwise, by the loop kernels beyond 16 KiB -- every boundary between those (and the padding
rule of choose_ld) is crossed here, for the three storage dtypes, against numpy in f64 on what
the index really stores."""
import numpy as np
import pytest

from oracle import svs_oracle as oracle
from synth import corpus_and_query

pytestmark = pytest.mark.gpu

DIMS = [1, 2, 3, 4, 5, 15, 16, 17, 31, 33, 63, 64, 65, 100, 127, 129, 200, 255, 256, 257, 300, 384, 500, 511, 513,
        640, 768, 1000, 1023, 1025, 1280, 1500, 1537, 2000, 2049, 2500, 3000, 3073, 4097, 5000]


@pytest.mark.parametrize("dtype", ["f32", "f16", "fp8"])
def test_scores_at_many_dimensions(gpu, dtype):
    from svs_amd import DeviceIndex
    tol = {"f32": 2e-6, "f16": 2e-6, "fp8": 5e-6}[dtype]
    for d in DIMS:
        n = 1237 if d > 1024 else 4099          # odd sizes: partial last groups, partial last wave
        m, qs = corpus_and_query("gaussian", 600 + d, n, d, 2)
        idx = DeviceIndex(m, dtype=dtype)
        assert idx.shape == (n, d) and idx.ld >= d
        md = m if dtype == "f32" else idx.stored_rows()
        for q in qs:
            qd = q if dtype == "f32" else idx.stored_query(q)
            got = idx.scores(q)
            want = oracle.cpu_scores_f64(md, qd)
            assert got.shape == (n,)
            err = np.max(np.abs(got.astype(np.float64) - want))
            assert err <= tol, f"{dtype} d={d} ld={idx.ld}: max |score - f64| = {err}"
        # and the whole search agrees with the total-order rule on those scores
        top = idx.search(qs[0], 10)
        assert [i for _, i in top] == [i for _, i in oracle.total_order_top_k(idx.scores(qs[0]), 10)]
        idx.release()


def test_row_stride_rule(gpu):
    """choose_ld: whole 1 KiB wave loads, else whole 128-byte lines, when that costs at most an
    eighth more bytes; always whole 16 bytes."""
    from svs_amd import DeviceIndex
    m = np.zeros((8, 1), dtype=np.float32)
    for dtype, per16, cases in (("f32", 4, {1000: 1024, 1536: 1536, 384: 384, 1500: 1536, 100: 100, 250: 256, 1100: 1120, 3: 4}),
                                ("f16", 8, {1000: 1024, 768: 768, 384: 384, 100: 104, 1537: 1600}),
                                ("fp8", 16, {1000: 1024, 1536: 1536, 384: 384, 100: 112, 3072: 3072})):
        for d, ld in cases.items():
            idx = DeviceIndex(np.zeros((8, d), dtype=np.float32), dtype=dtype)
            assert idx.ld == ld and idx.ld % per16 == 0, (dtype, d, idx.ld, ld)
            idx.release()
