import sys, os, faulthandler
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from synth import corpus_and_query
from svs_amd import DeviceIndex
n, d, k = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
m, _ = corpus_and_query("gaussian", 900 + n, n, d, 1)
m[n // 2] = m[7]; m[n - 1] = m[7]; m[n - 2] = m[n // 3]
idx = DeviceIndex(m, dtype=sys.argv[1] if len(sys.argv) > 1 else "f32")
ref = idx.top_pairs(k)
print("ref ok", ref[:2], flush=True)
idx.set_variant(1)
got = idx.top_pairs(k)
print("tiled ok", got[:2], got == ref, flush=True)
idx.set_variant(0)
idx.mask_rows([7, n - 2])
ref = idx.top_pairs(k)
idx.set_variant(1)
got = idx.top_pairs(k)
print("masked ok", got == ref, flush=True)
