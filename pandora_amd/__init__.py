"""pandora_amd: the read -> PRG minimizer path of `pandora map` on MI355X (HIP, gfx950).

The product is libpandora_amd.so (C ABI in include/pandora_amd.h); `pandora_amd.api` is a thin
ctypes face over it for tests and benchmarks.
"""
from . import api  # noqa: F401
from .api import (Index, MapResult, Pangraph, PmapError, Reads, filter_clusters, fraction, init,  # noqa: F401
                  map_reads, pangraph_from_read_file, read_fastaq)
