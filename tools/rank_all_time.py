#!/usr/bin/env python3
"""Full-ranking calls (k > 2048: the global-sort path of run_select), timed at the C-ABI boundary (tools only).
The reference's Dad-Jokes notebook ranks ALL 10,548 docs for one query in 36 ms
(examples/dad_jokes/Build Dad Jokes KB.ipynb:171-173: retrieve(query, n = len(docs))).
usage: rank_all_time.py [cases "n:k,n:k,..."] [d=1536]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from svs_amd import DeviceIndex

cases = [tuple(int(x) for x in c.split(":")) for c in (sys.argv[1] if len(sys.argv) > 1 else "10548:10548,10548:2048,65536:65536,1000000:10000,1000000:100").split(",")]
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
built = {}
for n, k in cases:
    if n not in built:
        for v in built.values():
            v.release()
        built.clear()
        torch.cuda.empty_cache()
        m = torch.randn((n, d), device=dev, generator=g); m /= m.norm(dim=1, keepdim=True)
        built[n] = DeviceIndex.from_device_pointer(m.data_ptr(), n, d, device=0); del m
    idx = built[n]
    qs = torch.randn((40, d), device=dev, generator=g); qs /= qs.norm(dim=1, keepdim=True)
    qh = qs.cpu().numpy()
    for q in qh[:8]:
        idx.search_batch(q[None, :], k)
    lat = []
    idx.set_timing(True)
    for q in qh[8:]:
        t0 = time.perf_counter()
        r = idx.search_batch(q[None, :], k)
        lat.append((time.perf_counter() - t0) * 1e3)
    sc, sel, cnt = idx.get_timing(); idx.set_timing(False)
    print(f"f32 {n}x{d} k={k}: p50 {np.median(lat):8.3f} ms  min {min(lat):8.3f} ms at the host API | device stages: score {sc/cnt:7.3f} ms, select {sel/cnt:7.3f} ms", flush=True)
for v in built.values():
    v.release()
