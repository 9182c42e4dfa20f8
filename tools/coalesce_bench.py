#!/usr/bin/env python3
"""Throughput of CONCURRENT single-query searches (what a server's AsyncKB.retrieve tasks do: one
executor thread each, reference src/svs/kb.py:1184-1190) with and without the search coalescer.
usage: coalesce_bench.py [n=1000000] [d=1536] [dtype=f32] [threads=1,4,16,64] [seconds=3] [modes=solo,coalesced,native]
(SVS_AMD_QUICK=0: the bookkeeping calls of a search through the GIL-releasing binding as well -- svs_amd/_native.py QUICK)"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from svs_amd import DeviceIndex
from svs_amd.coalesce import SearchCoalescer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
threads = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "1,4,16,64").split(",")]
secs = float(sys.argv[5]) if len(sys.argv) > 5 else 3.0
modes = (sys.argv[6] if len(sys.argv) > 6 else "solo,coalesced,native").split(",")
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(11)
idx = DeviceIndex.empty(d, device=0, dtype=dtype, reserve=n)
for r0 in range(0, n, 250000):
    m = torch.randn((min(n, r0 + 250000) - r0, d), device=dev, generator=g); m /= m.norm(dim=1, keepdim=True)
    idx.append_device(m.data_ptr(), m.shape[0]); del m
qs = torch.randn((256, d), device=dev, generator=g); qs /= qs.norm(dim=1, keepdim=True)
qh = qs.cpu().numpy()
for T in threads:
    for mode in modes:
        co = SearchCoalescer() if mode == "coalesced" else None
        idx.set_coalesce(mode == "native")
        p0, q0 = idx.coalesce_stats()
        done, lat, stop = [0] * T, [[] for _ in range(T)], time.time() + secs
        def worker(t):
            i = t
            while time.time() < stop:
                a = time.perf_counter()
                if co is not None: co.search(idx, qh[i % 256], 100)
                else: idx.search(qh[i % 256], 100)
                lat[t].append(time.perf_counter() - a)
                i += T; done[t] += 1
        ts = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
        t0 = time.time(); [t.start() for t in ts]; [t.join() for t in ts]; dt = time.time() - t0
        all_lat = np.sort(np.concatenate([np.array(x) for x in lat])) * 1e3
        extra = f", {co.queries / max(co.batches, 1):.1f} queries per corpus pass" if co is not None else ""
        if mode == "native":
            p1, q1 = idx.coalesce_stats()
            extra = f", {(q1 - q0) / max(p1 - p0, 1):.1f} queries per corpus pass (inside the library)"
        print(f"{T:3d} threads {mode:9s}: {sum(done) / dt:9.0f} queries/s, latency p50 {all_lat[len(all_lat) // 2]:.2f} ms p99 {all_lat[int(len(all_lat) * 0.99)]:.2f} ms{extra}", flush=True)
idx.release()
