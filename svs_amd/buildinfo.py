"""Identity of the kernel sources a measurement belongs to.

``csrc_sha16()`` hashes the CONTENT of svs_amd/csrc/* and include/svs_amd.h (sorted by name), not
a commit: a profile stays valid across documentation-only commits and goes stale with the first
change to a kernel or to the host code that launches it.  tools/profile_round.sh stamps every
rocprofv3 pass with it on the GPU box (which has no .git); bench.py quotes PMC traffic from
profiles/ only when the stamp equals the tree it is running from.
"""
from __future__ import annotations

import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_files():
    d = os.path.join(ROOT, "svs_amd", "csrc")
    out = [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith((".h", ".hip", ".cpp")) or f == "Makefile"]
    out.append(os.path.join(ROOT, "include", "svs_amd.h"))
    return out


def csrc_sha16() -> str:
    h = hashlib.sha256()
    for p in csrc_files():
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_sha16())
