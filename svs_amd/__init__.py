"""
svs_amd -- MI355X (gfx950) brute-force similarity backend for SVS.

Replaces the one hot path of Rhobota/svs: ``np.dot(embeddings_matrix, q)`` +
``get_top_k`` inside ``KB.retrieve()`` / ``AsyncKB.retrieve()`` (reference
src/svs/kb.py:1622-1627, src/svs/util.py:190-203) with hand-written HIP kernels
behind a C ABI (include/svs_amd.h).  See DESIGN.md and INTEGRATION.md.
"""
from ._native import device_count, lib_path  # noqa: F401
from .index import DeviceIndex  # noqa: F401
from .kb import KB, AsyncKB  # noqa: F401
from .matrix import DeviceEmbeddingsMatrix, attach  # noqa: F401
from .multi import MultiDeviceIndex  # noqa: F401

__version__ = "0.3.0"
