"""cbird_amd -- MI355X-native perceptual-hash build + Hamming nearest-neighbour find for cbird.

Only the hot path: HIP kernels + C-ABI in ``csrc/`` / ``include/cbird_hip.h`` and
this thin host mirror of the reference's Index plugin surface.  Importing the package does not
load the shared library; the first call does, and fails loudly when it (or a gfx950 device) is
missing.
"""
from ._lib import CbhError, lib, require_device  # noqa: F401
from .hashing import dct_hash64, dct_hash64_batch  # noqa: F401
from .index import DctFeaturesIndex, DctHashIndex, Match, MatchRange, Media, SearchParams  # noqa: F401

__all__ = ["CbhError", "lib", "require_device", "dct_hash64", "dct_hash64_batch", "DctHashIndex",
           "DctFeaturesIndex",
           "Match", "MatchRange", "Media", "SearchParams"]
