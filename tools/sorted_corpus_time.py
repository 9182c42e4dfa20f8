#!/usr/bin/env python3
"""A corpus SORTED by similarity to the queries (cosine to a direction u rises with the row index): with thresholds from
the FIRST rows (rounds 1-3; `svs_internal_tune(2, 0)`, pass `first` as the fifth argument) they cut nothing, every candidate
list of the fused batch path overflows and the host entry re-runs the queries through the materialised path; with the
sample spread over the corpus (round 4's default) nothing overflows.  Prints the call time beside an ordinary batch of the
same shape on the same index.  usage: sorted_corpus_time.py [n=1000000] [d=1536] [dtype=f16] [nq=1024] [first|spread]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from svs_amd import DeviceIndex, _native

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
dtype = sys.argv[3] if len(sys.argv) > 3 else "f16"
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
layout = sys.argv[5] if len(sys.argv) > 5 else "spread"
_native.load().svs_internal_tune(2, 0 if layout == "first" else 1)
k = 100
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(3)
u = torch.randn(d, device=dev, generator=g)
u /= u.norm()
idx = DeviceIndex.empty(d, device=0, dtype=dtype, reserve=n)
for r0 in range(0, n, 250_000):
    rows = min(250_000, n - r0)
    v = torch.randn((rows, d), device=dev, generator=g)
    v -= (v @ u)[:, None] * u[None, :]
    v /= v.norm(dim=1, keepdim=True)
    c = torch.linspace(0.05 + 0.9 * r0 / n, 0.05 + 0.9 * (r0 + rows) / n, rows, device=dev)[:, None]
    m = c * u[None, :] + torch.sqrt(1 - c * c) * v
    idx.append_device(m.contiguous().data_ptr(), rows)
    del v, m
qa = u[None, :] + 0.02 * torch.randn((nq, d), device=dev, generator=g)
qa = (qa / qa.norm(dim=1, keepdim=True)).cpu().numpy()
qr = torch.randn((nq, d), device=dev, generator=g)
qr = (qr / qr.norm(dim=1, keepdim=True)).cpu().numpy()
ph = (C.c_double * 6)()
lib = _native.load()
print(f"thresholds from: {layout} rows")
for name, q in (("queries along u", qa), ("random queries", qr)):
    idx.search_batch(q, k)
    t = []
    for _ in range(3):
        a = time.perf_counter()
        s, r = idx.search_batch(q, k)
        t.append(time.perf_counter() - a)
    lib.svs_internal_host_phases(ph, 6)
    print(f"{name}: {sorted(t)[1] * 1e3:.2f} ms per call of {nq}, {int(ph[5])} queries re-run; best row of query 0: {int(r[0, 0])} (score {float(s[0, 0]):.4f})", flush=True)
idx.release()
