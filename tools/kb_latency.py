#!/usr/bin/env python3
"""End-to-end KB.retrieve() latency incl. SQLite (SURVEY.md 8(d) 'Latency') and the
cold-start matrix build (8(f) rank 1), on a Dad-Jokes-sized synthetic KB
(BASELINE.json configs[0]: 10,548 docs x 1536).  Run on the GPU box."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import svs_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10548
d = 1536
rng = np.random.default_rng(0)
vecs = rng.standard_normal((n + 64, d)); vecs /= np.linalg.norm(vecs, axis=1, keepdims=True)
vecs = vecs.astype(np.float32)
lookup = {f"doc {i}": i for i in range(n + 64)}

async def ef(texts):
    return [vecs[lookup[t]].tolist() for t in texts]

with tempfile.TemporaryDirectory() as td:
    kb = svs_amd.KB(os.path.join(td, "kb.sqlite"), ef)
    t0 = time.perf_counter()
    with kb.bulk_add_docs() as add_doc:
        for i in range(n):
            add_doc(f"doc {i}")
    print(f"bulk add of {n} docs: {time.perf_counter()-t0:.2f} s")
    t0 = time.perf_counter()
    kb.load()
    print(f"cold start (SQLite -> matrix -> HBM): {time.perf_counter()-t0:.3f} s for {n} rows")
    lat = []
    for i in range(n, n + 64):
        t0 = time.perf_counter(); docs = kb.retrieve(f"doc {i}", 100); lat.append(time.perf_counter() - t0)
    print(f"KB.retrieve(n=100) p50 {np.median(lat)*1e3:.3f} ms  min {min(lat)*1e3:.3f} ms (embed lookup + HIP search + SQLite fetch)")
    qs = [f"doc {i}" for i in range(n, n + 64)]
    t0 = time.perf_counter(); many = kb.retrieve_many(qs, 100); dt = time.perf_counter() - t0
    print(f"KB.retrieve_many(64 queries, n=100): {dt*1e3:.2f} ms total = {dt/64*1e3:.3f} ms/query")
    assert [d_["doc"]["id"] for d_ in many[0]] == [d_["doc"]["id"] for d_ in kb.retrieve(qs[0], 100)]
    kb.close()
