#!/usr/bin/env python3
"""document_top_pairwise_scores at a size the reference cannot touch (its n x n f32 matrix would be
4 TB at 1M rows): time svs_index_top_pairs on a device-generated corpus with planted near-duplicates.
usage: pairs_time.py [n=1000000] [d=1536] [dtype=f16] [k=100]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svs_amd import DeviceIndex
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
dtype = sys.argv[3] if len(sys.argv) > 3 else "f16"
k = int(sys.argv[4]) if len(sys.argv) > 4 else 100
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(7)
idx = DeviceIndex.empty(d, device=0, dtype=dtype, reserve=n)
planted = []
blk = 250_000
for r0 in range(0, n, blk):
    m = torch.randn((min(n, r0 + blk) - r0, d), device=dev, generator=g); m /= m.norm(dim=1, keepdim=True)
    for t in range(5):                      # five near-duplicate pairs per block, at known rows
        a, b = 1000 * (t + 1), m.shape[0] - 1000 * (t + 1)
        m[b] = m[a] + 0.01 * (t + 1) * torch.randn(d, device=dev, generator=g) / d ** 0.5
        m[b] /= m[b].norm()
        planted.append((r0 + a, r0 + b))
    idx.append_device(m.data_ptr(), m.shape[0])
    del m
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    got = idx.top_pairs(k)
    dt = time.perf_counter() - t0
    flop = float(n) * n * d          # n^2 / 2 pairs x 2 d flop
    print("top_pairs(%d) over %d x %d %s: %.3f s  (%.0f TFLOP/s over the upper triangle)" % (k, n, d, dtype, dt, flop / dt / 1e12), flush=True)
top = {(i, j) for _, i, j in got[:len(planted)]}
print("planted pairs found in the top %d: %d of %d;  best %s" % (len(planted), len(top & set(planted)), len(planted), got[0]))
idx.release()
