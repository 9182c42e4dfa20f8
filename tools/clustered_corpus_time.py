#!/usr/bin/env python3
"""A corpus stored TOPIC BY TOPIC (`clusters` Gaussian clusters around random unit centres, one after the other) and
queries drawn from random topics: the fused batch path with its thresholds taken from the first rows (rounds 1-3: all of
one topic) against the sample spread over the corpus (round 4).  usage: clustered_corpus_time.py [n=1000000] [d=1536]
[dtype=f16] [nq=1024] [clusters=32] [spread=0.8]"""
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
ncl = int(sys.argv[5]) if len(sys.argv) > 5 else 32
spread = float(sys.argv[6]) if len(sys.argv) > 6 else 0.8
k = 100
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(5)
centres = torch.randn((ncl, d), device=dev, generator=g)
centres /= centres.norm(dim=1, keepdim=True)
per = (n + ncl - 1) // ncl
idx = DeviceIndex.empty(d, device=0, dtype=dtype, reserve=n)
for c in range(ncl):
    rows = min(per, n - c * per)
    if rows <= 0:
        break
    m = centres[c][None, :] + spread * torch.randn((rows, d), device=dev, generator=g) / d ** 0.5
    m /= m.norm(dim=1, keepdim=True)
    idx.append_device(m.contiguous().data_ptr(), rows)
    del m
topic = torch.randint(0, ncl, (nq,), device=dev, generator=g)
q = centres[topic] + spread * torch.randn((nq, d), device=dev, generator=g) / d ** 0.5
q = (q / q.norm(dim=1, keepdim=True)).cpu().numpy()
ph = (C.c_double * 6)()
lib = _native.load()
res = {}
if os.environ.get("CCT_DIV"):              # threshold sample of n / CCT_DIV rows (default 64)
    lib.svs_internal_tune(0, int(os.environ["CCT_DIV"]))
if os.environ.get("CCT_VARIANT"):          # e.g. CCT_VARIANT=6: the materialised path (no fused epilogue) on the same data
    idx.set_variant(int(os.environ["CCT_VARIANT"]))
for layout in ("first", "spread"):
    lib.svs_internal_tune(2, 0 if layout == "first" else 1)
    idx.search_batch(q, k)
    idx.set_timing(True)
    t = []
    for _ in range(5):
        a = time.perf_counter()
        s, r = idx.search_batch(q, k)
        t.append(time.perf_counter() - a)
    sc, se, cnt = idx.get_timing()
    idx.set_timing(False)
    lib.svs_internal_host_phases(ph, 6)
    res[layout] = r
    print(f"thresholds from {layout:6s} rows: {sorted(t)[2] * 1e3:.2f} ms per call of {nq} ({ncl} topics, {n} x {d} {dtype}); "
          f"score stage {sc / max(cnt, 1):.2f} ms, select stage {se / max(cnt, 1):.3f} ms, {int(ph[5])} queries re-run", flush=True)
lib.svs_internal_tune(2, 1)
print("same rows either way:", bool((res["first"] == res["spread"]).all()))
idx.release()
