#!/usr/bin/env python3
"""Throughput table over (dtype, queries per call) on one GPU (tools only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from svs_amd import DeviceIndex

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
m = torch.randn((n, d), device=dev, generator=g); m /= m.norm(dim=1, keepdim=True)
for dtype in ("f32", "f16"):
    idx = DeviceIndex.from_device_pointer(m.data_ptr(), n, d, device=0, dtype=dtype)
    esz = 4 if dtype == "f32" else 2
    for nq in (1, 16, 32, 64, 256, 1024):
        qs = torch.randn((nq, d), device=dev, generator=g); qs /= qs.norm(dim=1, keepdim=True)
        qh = qs.cpu().numpy()
        idx.search_batch(qh, 100)
        reps = max(2, min(20, 64 // nq))
        idx.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            idx.search_batch(qh, 100)
        wall = time.perf_counter() - t0
        sc, sel, cnt = idx.get_timing(); idx.set_timing(False)
        per = {"f32": 16, "f16": 32}[dtype] if nq > 1 else 1
        passes = (nq + per - 1) // per
        print(f"{dtype} nq={nq:5d}: score {sc/cnt:9.3f} ms ({n*d*esz*passes/(sc/cnt*1e-3)/1e12:5.2f} TB/s over {passes} passes) "
              f"select {sel/cnt:8.3f} ms  -> {nq*reps/wall:10.1f} qps", flush=True)
    idx.release()
