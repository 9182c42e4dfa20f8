#!/usr/bin/env python3
"""Cold start of a KB at the reference benchmark's size (tools only): SQLite file -> first result.
The reference publishes 98.7 s for the matrix build alone and 1 min 38 s for the first query at
1M x 1536 (examples/One Million Documents Benchmark.ipynb:236-248).
usage: kb_coldstart.py [N=1000000] [D=1536] [path=/tmp/svs_cold.sqlite]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import svs_amd
from svs_amd.kb import _Store

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
path = sys.argv[3] if len(sys.argv) > 3 else "/tmp/svs_cold.sqlite"
rng = np.random.default_rng(1)
qv = rng.standard_normal(d); qv /= np.linalg.norm(qv)

async def ef(texts):
    return [[float(x) for x in qv] for _ in texts]

if os.path.exists(path):
    os.remove(path)
t0 = time.perf_counter()
st = _Store(path)
blk = 20000
with st.transaction():
    for b0 in range(0, n, blk):
        x = rng.standard_normal((min(blk, n - b0), d)).astype(np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        st.conn.executemany("INSERT INTO embeddings (embedding) VALUES (?)", [(r.tobytes(),) for r in x])
        st.conn.executemany("INSERT INTO docs (parent_id, level, text, embedding, meta) VALUES (NULL, 0, ?, ?, NULL)",
                            [(f"doc {b0 + i}", b0 + i + 1) for i in range(len(x))])
st.close()
print(f"wrote {n} x {d} KB ({os.path.getsize(path) / 1e9:.2f} GB) in {time.perf_counter() - t0:.1f} s", flush=True)

for mode in ("BLOBs -> pinned staging blocks -> HBM (svs_index_staging_*)", "matrix first (round 1)"):
    os.system("sync")
    kb = svs_amd.KB(path, ef)
    if mode.startswith("matrix"):
        kb.embeddings_matrix._block_builder = None
    t0 = time.perf_counter()
    kb.load()
    t1 = time.perf_counter()
    res = kb.retrieve("q", 100)
    t2 = time.perf_counter()
    res = kb.retrieve("q", 100)
    t3 = time.perf_counter()
    import resource
    print(f"{mode}: cold matrix build + upload {t1 - t0:.2f} s, first retrieve {1e3 * (t2 - t1):.2f} ms, next {1e3 * (t3 - t2):.2f} ms; "
          f"first query wall {t2 - t0:.2f} s; peak RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6:.2f} GB; top score {res[0]['score']:.4f}", flush=True)
    kb.close()
os.remove(path)
