"""coltt_amd — MI355X-native ANN search hot path of sjy-dv/coltt.

The product is libcoltt_gpu.so (hand-written HIP for gfx950 behind the C-ABI of include/coltt_gpu.h).
This package is the thin ctypes binding used by tests/ and bench.py, shaped after the reference's Go
interfaces (edge.vectorspace: edge/vectorstore.go:30-49; *vectorindex.Hnsw: core/vectorindex/hnsw.go).
There is NO CPU fallback: importing works without a GPU (so the symbol table can be checked), but every
compute entry point fails loudly when the extension or the device is missing.
"""
from ._lib import (COSINE, EUCLIDEAN, Q_NONE, Q_F16, Q_F8, Q_BF16, SELECT_REFERENCE, SELECT_NEAREST,  # noqa: F401
                   MODE_EXACT, MODE_MFMA, ColttError, lib, lib_path, declared_symbols)
from .flat import FlatSpace  # noqa: F401
from .hnsw import Hnsw, HnswCfg  # noqa: F401
from .cflat import MultiVectorSpace  # noqa: F401
from .pq import PQSpace, PQ_COSINE, PQ_EUCLIDEAN, PQ_DOT  # noqa: F401
from . import kernels  # noqa: F401
from . import group  # noqa: F401
from .group import Group  # noqa: F401
