#!/usr/bin/env python3
"""Quick A/B of score-stage variants on the GPU box (not part of the product)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from svs_amd import DeviceIndex

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1234)
m = torch.randn((n, d), device=dev, dtype=torch.float32, generator=g)
m /= m.norm(dim=1, keepdim=True)
q = torch.randn((d,), device=dev, generator=g); q /= q.norm()
torch.cuda.synchronize()
idx = DeviceIndex.from_device_pointer(m.data_ptr(), n, d, device=0)
del m
torch.cuda.empty_cache()
qh = q.cpu().numpy()
bytes_q = n * d * 4
for variant in (0,):
    idx.set_variant(variant)
    for _ in range(3):
        idx.search(qh, 100)
    idx.set_timing(True)
    t0 = time.perf_counter()
    lat = []
    for _ in range(iters):
        t1 = time.perf_counter()
        idx.search(qh, 100)
        lat.append(time.perf_counter() - t1)
    wall = time.perf_counter() - t0
    sc, sel, cnt = idx.get_timing()
    idx.set_timing(False)
    print(f"variant {variant}: score {sc/cnt*1e3:8.1f} us  ({bytes_q/(sc/cnt*1e-3)/1e12:5.2f} TB/s)  select {sel/cnt*1e3:7.1f} us  "
          f"p50 latency {np.median(lat)*1e3:7.3f} ms  qps(sync) {iters/wall:8.1f}", flush=True)

# ---- batched throughput (queries per corpus pass)
g2 = torch.Generator(device=dev); g2.manual_seed(5)
for variant in (0, 3, 4, 5):
  idx.set_variant(variant)
  for nq in (16,):
    qs = torch.randn((nq, d), device=dev, generator=g2); qs /= qs.norm(dim=1, keepdim=True)
    qsh = qs.cpu().numpy()
    for _ in range(2):
        idx.search_batch(qsh, 100)
    idx.set_timing(True)
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        idx.search_batch(qsh, 100)
    wall = time.perf_counter() - t0
    sc, sel, cnt = idx.get_timing()
    idx.set_timing(False)
    print(f"variant {variant} batch {nq:3d}: score {sc/cnt*1e3:8.1f} us  select {sel/cnt*1e3:7.1f} us  -> {nq*reps/wall:9.1f} qps "
          f"(corpus bytes/s per pass {bytes_q*((nq+15)//16)/(sc/cnt*1e-3)/1e12:5.2f} TB/s)", flush=True)
