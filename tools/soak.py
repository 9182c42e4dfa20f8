#!/usr/bin/env python3
"""Soak (tools only): several threads hammer ONE index handle with a mix of single, batched
fused and batched materialised searches (all dtypes in turn) and compare every answer with
the answer the same call gave when it ran alone.  usage: soak.py [seconds_per_dtype]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from svs_amd import DeviceIndex

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(5)
n, d = 400_000, 1536
m = torch.randn((n, d), device=dev, generator=g); m /= m.norm(dim=1, keepdim=True)
qs = torch.randn((300, d), device=dev, generator=g); qs /= qs.norm(dim=1, keepdim=True)
qh = qs.cpu().numpy()
for dtype in ("f32", "f16", "fp8"):
    idx = DeviceIndex.from_device_pointer(m.data_ptr(), n, d, device=0, dtype=dtype)
    jobs = [qh[0:1], qh[1:17], qh[17:24], qh[24:64], qh[64:81], qh[0:300], qh[100:228]]
    want = [idx.search_batch(q, 100) for q in jobs]
    bad, calls, stop = [], [0], time.time() + secs
    def worker(t):
        it = 0
        while time.time() < stop and not bad:
            j = (t * 3 + it) % len(jobs)
            s, r = idx.search_batch(jobs[j], 100)
            if not (np.array_equal(r, want[j][1]) and np.array_equal(s, want[j][0])):
                bad.append((t, it, j))
            it += 1
            calls[0] += 1
    ts = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    [t.start() for t in ts]; [t.join() for t in ts]
    print(f"{dtype}: {calls[0]} calls, mismatches: {bad}", flush=True)
    idx.release()
    if bad:
        sys.exit(1)
print("soak ok")
