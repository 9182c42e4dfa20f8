#!/usr/bin/env python3
"""Throughput table over (dtype, queries per call) on one GPU (tools only).
usage: perf_matrix.py N D dtypes nqs   e.g.  perf_matrix.py 10000000 3072 fp8 1,256"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from svs_amd import DeviceIndex

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
dtypes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["f32", "f16", "fp8"]
nqs = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 16, 32, 64, 256, 1024]
variant = int(sys.argv[5]) if len(sys.argv) > 5 else 0
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
blk = 500_000
esz = {"f32": 4, "f16": 2, "fp8": 1}
for dtype in dtypes:
    # build in blocks so the f32 source of a large corpus need not exist at once
    if n <= 2_000_000:
        m = torch.randn((n, d), device=dev, generator=g); m /= m.norm(dim=1, keepdim=True)
        idx = DeviceIndex.from_device_pointer(m.data_ptr(), n, d, device=0, dtype=dtype); del m
        shards = [idx]
    else:
        shards = []
        for r0 in range(0, n, blk):
            r1 = min(n, r0 + blk)
            m = torch.randn((r1 - r0, d), device=dev, generator=g); m /= m.norm(dim=1, keepdim=True)
            shards.append(m)
        big = torch.cat(shards); del shards, m
        idx = DeviceIndex.from_device_pointer(big.data_ptr(), n, d, device=0, dtype=dtype); del big
    torch.cuda.empty_cache()
    idx.set_variant(variant)
    for nq in nqs:
        qs = torch.randn((nq, d), device=dev, generator=g); qs /= qs.norm(dim=1, keepdim=True)
        qh = qs.cpu().numpy()
        for _ in range(max(1, min(12, 192 // nq))):   # past the clock's settling time after a change of kernel mix
            idx.search_batch(qh, 100)
        reps = max(2, min(20, 256 // nq))
        idx.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            idx.search_batch(qh, 100)
        wall = time.perf_counter() - t0
        sc, sel, cnt = idx.get_timing(); idx.set_timing(False)
        flops = 2.0 * n * d * nq
        print(f"{dtype} {n}x{d} nq={nq:5d}: score {sc/cnt:9.3f} ms  select {sel/cnt:8.3f} ms -> {nq*reps/wall:10.1f} qps | "
              f"one corpus pass/call would be {n*d*esz[dtype]/(sc/cnt*1e-3)/1e12:6.2f} TB/s | {flops/(sc/cnt*1e-3)/1e12:7.1f} TFLOP/s", flush=True)
    idx.release()
    torch.cuda.empty_cache()
