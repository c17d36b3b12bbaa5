"""taxor_amd -- MI355X-native `taxor search` hot path (syncmers -> HIXF query -> per-bin tally).

The product is libtaxor_gpu.so (hand-written HIP for gfx950 behind the C ABI in include/taxor_gpu.h) plus the
C++ `taxor search` host (taxor_amd/csrc).  This package is the Python mirror of the host interface used by
the tests and bench.py.  There is no CPU fallback: without the HIP library every compute call raises."""
from . import _lib
from .search import Comm, GpuIndex, Searcher, SearchResults, classify_filter, threshold, threshold_ratio  # noqa: F401

__all__ = ["Comm", "GpuIndex", "Searcher", "SearchResults", "classify_filter", "threshold", "threshold_ratio", "_lib"]
