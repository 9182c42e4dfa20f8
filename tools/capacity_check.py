#!/usr/bin/env python3
"""One GPU filled to most of its 288 GB: an f16 index of `rows` x 1536 built block by block ON the device (no host
matrix), then single-query searches whose answers are checked against the full score vector of the same index
(svs_index_scores_n -> host argpartition: the reference's own get_top_k on np.dot's output, src/svs/util.py:190-203)
and against rows planted at known positions, the last one at the very end of the HBM image.
usage: capacity_check.py [rows=80000000] [d=1536] [dtype=f16] [queries=3]
(80M x 1536 f16 = 245.8 GB; the f32 blocks the rows are generated in are 1M rows = 6.1 GB and are freed as they go.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from svs_amd import DeviceIndex

n = int(sys.argv[1]) if len(sys.argv) > 1 else 80_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
dtype = sys.argv[3] if len(sys.argv) > 3 else "f16"
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 3
k = 100
dev = torch.device("cuda:0")
free, total = torch.cuda.mem_get_info(dev)
esz = {"f32": 4, "f16": 2, "fp8": 1}[dtype]
need = n * d * esz + (8 << 30)
print(f"HBM: {free / 2**30:.1f} GiB free of {total / 2**30:.1f}; the index needs {n * d * esz / 1e9:.1f} GB", flush=True)
if need > free:
    sys.exit(f"not enough free HBM for {n} rows (would need {need / 2**30:.1f} GiB with the working space)")

g = torch.Generator(device=dev)
g.manual_seed(2024)
qs = torch.randn((nq, d), device=dev, generator=g)
qs /= qs.norm(dim=1, keepdim=True)
# planted rows: query j itself (score ~1) at these positions -- the first row, an odd row in the middle (byte offset > 2^36), the last row
plant = {0: 0, (n // 2) | 1: 1 % nq, n - 1: 2 % nq}
block = 1_000_000
t0 = time.time()
idx = DeviceIndex.empty(d, device=0, dtype=dtype, reserve=n)
for r0 in range(0, n, block):
    rows = min(block, n - r0)
    m = torch.randn((rows, d), device=dev, generator=g)
    m /= m.norm(dim=1, keepdim=True)
    for pos, j in plant.items():
        if r0 <= pos < r0 + rows:
            m[pos - r0] = qs[j]
    idx.append_device(m.data_ptr(), rows)
    del m
    if (r0 // block) % 10 == 9:
        print(f"  {r0 + rows:,} rows in HBM ({time.time() - t0:.0f} s)", flush=True)
torch.cuda.synchronize()
print(f"built {idx.n:,} x {d} {dtype} = {idx.hbm_bytes / 1e9:.1f} GB in {time.time() - t0:.0f} s", flush=True)

qh = qs.cpu().numpy()
bad = 0
for j in range(nq):
    res = idx.search(qh[j], k)
    t = []
    for _ in range(5):
        a = time.perf_counter()
        idx.search(qh[j], k)
        t.append(time.perf_counter() - a)
    ms = sorted(t)[len(t) // 2] * 1e3
    sc = idx.scores(qh[j])                                   # the whole np.dot(M, q) vector of THIS index, f32 (n)
    part = np.argpartition(sc, -k)[-k:]                      # src/svs/util.py:200-203
    want = sorted(((float(sc[i]), int(i)) for i in part), reverse=True)
    rows_ok = [r for _, r in res] == [r for _, r in want]
    score_ok = max(abs(a - b) for (a, _), (b, _) in zip(res, want)) <= 1e-5
    planted = [pos for pos, pj in plant.items() if pj == j]
    top = {r for _, r in res[:len(planted)]}
    plant_ok = set(planted) == top and all(abs(s - 1.0) < 2e-3 for s, _ in res[:len(planted)])
    ok = rows_ok and score_ok and plant_ok
    bad += not ok
    print(f"query {j}: {ms:.2f} ms per search = {idx.hbm_bytes / (ms * 1e-3) / 1e12:.2f} TB/s; rows equal to argpartition over the "
          f"score vector: {rows_ok}; scores within 1e-5: {score_ok}; planted rows {planted} on top: {plant_ok}", flush=True)
idx.release()
print("capacity check " + ("ok" if not bad else "FAILED"))
sys.exit(1 if bad else 0)
