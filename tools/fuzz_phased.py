#!/usr/bin/env python3
"""Fuzz (tools only): random panel shapes through the phased kernel (variant 0) and the round-1 tiled
kernel (variant 2) -- same arithmetic in the same k order, so scores and rows must agree bit for bit; each
shape is also searched three times (the repetitions must agree: the race screen for the LDS-DMA ring).
usage: fuzz_phased.py [cases=24] [seed=1]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from svs_amd import DeviceIndex
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
bad = 0
for c in range(cases):
    dtype = ["f16", "fp8"][int(rng.integers(2))]
    # row bytes must be an even number (>= 6) of 128-byte k-tiles for the phased kernel
    kt = int(rng.choice([6, 8, 10, 12, 14, 16, 20, 24, 32, 48]))
    d = kt * 128 // (2 if dtype == "f16" else 1)
    n = int(rng.integers(131_072, 420_000))
    nq = int(rng.choice([65, 66, 97, 127, 128, 129, 200, 255, 256, 257, 300, 511, 512, 700, 1024, 1100]))   # 65 .. 128: the 128-query tiles
    k = int(rng.choice([1, 3, 10, 100, 256]))   # (small k: few survivors per tile, the grouped epilogue path)
    if int(rng.integers(4)) == 0:
        n = int(rng.integers(3000, 60_000))          # materialised path (n < 131,072): the non-fused phased kernel
    g = torch.Generator(device=dev); g.manual_seed(1000 + c)
    m = torch.randn((n, d), device=dev, generator=g); m /= m.norm(dim=1, keepdim=True)
    idx = DeviceIndex.from_device_pointer(m.data_ptr(), n, d, device=0, dtype=dtype); del m
    q = torch.randn((nq, d), device=dev, generator=g); q = (q / q.norm(dim=1, keepdim=True)).cpu().numpy()
    s0, r0 = idx.search_batch(q, k)
    ok = True
    for rep in range(2):
        s1, r1 = idx.search_batch(q, k)
        ok &= bool(np.array_equal(s0, s1) and np.array_equal(r0, r1))
    idx.set_variant(2)
    s2, r2 = idx.search_batch(q, k)
    same = bool(np.array_equal(s0, s2) and np.array_equal(r0, r2))
    # brute-force spot check of three queries against torch on the stored rows
    md = torch.from_numpy(idx.stored_rows()).to(dev)
    spot = True
    for qi in (0, nq // 2, nq - 1):
        qd = torch.from_numpy(idx.stored_query(q[qi])).to(dev)
        sc = md @ qd
        top = torch.topk(sc, min(k, n)).values.cpu().numpy()
        spot &= bool(np.max(np.abs(top - s0[qi])) <= 2e-5)
    idx.release(); del md
    torch.cuda.empty_cache()
    print(f"case {c:2d}: {dtype} n={n} d={d} (k-tiles {kt}) nq={nq} k={k}: repeatable {ok}, == tiled {same}, scores vs torch {spot}", flush=True)
    bad += not (ok and same and spot)
print("fuzz", "ok" if not bad else f"FAILED ({bad} cases)")
sys.exit(1 if bad else 0)
