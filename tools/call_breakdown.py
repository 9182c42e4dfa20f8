#!/usr/bin/env python3
"""Where the time of a host-API batch call goes, under the library's internal knobs (svs_internal_tune):
upload = staged DMA (round 3) or the staging kernels pulling chunk by chunk from pinned memory,
the threshold prefix (n / div rows); and the throughput of TWO callers in flight on one handle.
usage: call_breakdown.py N D dtype nq [reps=12] [prefix divisors=64,128]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from svs_amd import DeviceIndex, _native

n, d, dtype, nq = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 12
divs = [int(x) for x in sys.argv[6].split(",")] if len(sys.argv) > 6 else [64, 128]
lib = _native.load()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
blk = 500_000
idx = DeviceIndex.empty(d, device=0, dtype=dtype, reserve=n)
for r0 in range(0, n, blk):
    m = torch.randn((min(n, r0 + blk) - r0, d), device=dev, generator=g); m /= m.norm(dim=1, keepdim=True)
    idx.append_device(m.data_ptr(), m.shape[0])
    del m
torch.cuda.empty_cache()
qs = torch.randn((nq, d), device=dev, generator=g); qs /= qs.norm(dim=1, keepdim=True)
qh = qs.cpu().numpy()
q2 = np.ascontiguousarray(qh[::-1])


def tune(what, v):
    _native.check(lib.svs_internal_tune(what, v))


def one(label):
    for _ in range(3):
        out = idx.search_batch(qh, 100)
    idx.set_timing(True)
    ts = []
    for _ in range(reps):
        a = time.perf_counter()
        out = idx.search_batch(qh, 100)
        ts.append((time.perf_counter() - a) * 1e3)
    score_ms, select_ms, cnt = idx.get_timing()
    dom = idx.last_dominant_ms_sum / max(cnt, 1)
    idx.set_timing(False)
    import ctypes as C
    ph = (C.c_double * 5)()
    lib.svs_internal_host_phases(ph, 5)
    print("%-40s call %.3f ms (min %.3f)  dominant kernel %.3f  score stage %.3f  select %.3f  -> outside the kernel %.3f ms | last call, host side (ms since entry): "
          "planned %.3f, queries in pinned memory %.3f, all enqueued %.3f, stream drained %.3f, results copied out %.3f"
          % (label, float(np.median(ts)), min(ts), dom, score_ms / cnt, select_ms / cnt, float(np.median(ts)) - dom, *[x * 1e3 for x in ph]), flush=True)
    return out


ref = None
for rd in range(2):
    for upload, div in [(1, 64), (0, 64)] + [(0, dv) for dv in divs if dv != 64]:
        tune(1, upload); tune(0, div)
        out = one("round %d upload=%s prefix=n/%d" % (rd, {0: "chunked pull", 1: "staged dma (r3)"}[upload], div))
        if ref is None:
            ref = out
        else:
            assert (out[1] == ref[1]).all() and (out[0] == ref[0]).all(), "results differ between knob settings"
tune(1, 0); tune(0, 64)

# two callers in flight (each its own thread, its own query batch; ctypes releases the GIL inside the call)
for callers in (1, 2, 3):
    stop = time.time() + 1.5
    done = [0] * callers

    def run(t):
        q = qh if t % 2 == 0 else q2
        while time.time() < stop:
            idx.search_batch(q, 100)
            done[t] += 1
    th = [threading.Thread(target=run, args=(t,)) for t in range(callers)]
    t0 = time.time()
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.time() - t0
    print("%d caller(s) in flight: %.0f queries/s (%.3f ms per call and caller)" % (callers, sum(done) * nq / dt, dt * callers / max(sum(done), 1) * 1e3), flush=True)
idx.release()
