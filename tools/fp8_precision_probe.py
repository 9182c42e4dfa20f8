import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from synth import corpus_and_query
from svs_amd import DeviceIndex
n, d, nq, k = 140_000, 256, 256, 100
m, qs = corpus_and_query("gaussian", 99, n, d, nq)
m = m * 0.5; m[:, 0] += 1.0; m /= np.linalg.norm(m, axis=1, keepdims=True)
qs = qs * 0.2; qs[:, 0] -= 1.0; qs /= np.linalg.norm(qs, axis=1, keepdims=True)
m, qs = m.astype(np.float32), qs.astype(np.float32)
idx = DeviceIndex(m, dtype="fp8")
s, r = idx.search_batch(qs, k)
md = idx.stored_rows(); qd = idx.stored_query(qs[0])
t64 = md.astype(np.float64) @ qd.astype(np.float64)
np32 = md @ qd
one = idx.search(qs[0], k)          # single-query kernel (f32 FMAs)
rows = r[0]
print("batch (MFMA)   vs f64: max |err| %.3e" % np.max(np.abs(s[0].astype(np.float64) - t64[rows])))
print("numpy f32      vs f64: max |err| %.3e" % np.max(np.abs(np32[rows].astype(np.float64) - t64[rows])))
print("single (FMA)   vs f64: max |err| %.3e" % np.max(np.abs(np.array([x for x,_ in one]) - t64[[i for _,i in one]])))
# the same on ordinary data
m2, q2 = corpus_and_query("gaussian", 5, n, d, nq)
i2 = DeviceIndex(m2, dtype="fp8"); s2, r2 = i2.search_batch(q2, k)
md2 = i2.stored_rows(); qd2 = i2.stored_query(q2[0]); t2 = md2.astype(np.float64) @ qd2.astype(np.float64)
print("ordinary data, batch vs f64: max |err| %.3e at scores ~%.3f" % (np.max(np.abs(s2[0].astype(np.float64) - t2[r2[0]])), s2[0][0]))
