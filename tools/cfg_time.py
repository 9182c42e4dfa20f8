#!/usr/bin/env python3
"""Dominant-kernel time of a batched search on REAL (unit-norm Gaussian) data, per kernel variant
(svs_index_set_variant), in one process: the A/B harness for the MFMA kernels.
usage: cfg_time.py N D dtype nq variants(comma) [reps=10] [rounds=3]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svs_amd import DeviceIndex
n, d, dtype, nq = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
variants = [int(v) for v in sys.argv[5].split(",")]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
rounds = int(sys.argv[7]) if len(sys.argv) > 7 else 3
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
blk = 500_000
idx = None
for r0 in range(0, n, blk):
    m = torch.randn((min(n, r0 + blk) - r0, d), device=dev, generator=g); m /= m.norm(dim=1, keepdim=True)
    if idx is None:
        idx = DeviceIndex.empty(d, device=0, dtype=dtype, reserve=n)
    idx.append_device(m.data_ptr(), m.shape[0])
    del m
torch.cuda.empty_cache()
qs = torch.randn((nq, d), device=dev, generator=g); qs /= qs.norm(dim=1, keepdim=True)
qh = qs.cpu().numpy()
ref = None
flop = 2.0 * n * d * nq
for rd in range(rounds):
    for v in variants:
        idx.set_variant(v)
        for _ in range(2):
            out = idx.search_batch(qh, 100)
        idx.set_timing(True)
        for _ in range(reps):
            out = idx.search_batch(qh, 100)
        score_ms, select_ms, cnt = idx.get_timing()
        dom = idx.last_dominant_ms_sum / max(cnt, 1)
        idx.set_timing(False)
        same = ""
        if ref is None:
            ref = out
        else:
            same = "  rows == first variant: %s" % bool((out[1] == ref[1]).all())
        print("round %d variant %d: dominant kernel %.3f ms (%.0f TFLOP/s), score stage %.3f ms, select %.3f ms%s"
              % (rd, v, dom, flop / dom / 1e9, score_ms / cnt, select_ms / cnt, same), flush=True)
idx.release()
