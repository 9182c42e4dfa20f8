#!/usr/bin/env python3
"""rocprofv3 target (tools only): N x D index, a few batched searches.
usage: prof_batch.py N D dtype nq [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svs_amd import DeviceIndex
n, d, dtype, nq = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
if n <= 2_000_000:
    m = torch.randn((n, d), device=dev, generator=g); m /= m.norm(dim=1, keepdim=True)
    idx = DeviceIndex.from_device_pointer(m.data_ptr(), n, d, device=0, dtype=dtype); del m
else:
    # built in blocks (as tools/perf_matrix.py): the f32 source exists only until the index is made
    blk, shards = 500_000, []
    for r0 in range(0, n, blk):
        m = torch.randn((min(n, r0 + blk) - r0, d), device=dev, generator=g); m /= m.norm(dim=1, keepdim=True)
        shards.append(m)
    big = torch.cat(shards); del shards, m
    idx = DeviceIndex.from_device_pointer(big.data_ptr(), n, d, device=0, dtype=dtype); del big
torch.cuda.empty_cache()
qs = torch.randn((nq, d), device=dev, generator=g); qs /= qs.norm(dim=1, keepdim=True)
qh = qs.cpu().numpy()
for _ in range(reps):
    idx.search_batch(qh, 100)
idx.release()
