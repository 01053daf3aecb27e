"""megapath_amd -- MI355X-native soap4 hot path (2BWT MMP seeding + semi-global DP) behind a C ABI.

The compute lives in libmegapath_hip.so (hand-written HIP for gfx950, megapath_amd/csrc/).  This package is the
thin Python mirror of that ABI plus host-side helpers (read packing, synthetic data, index building).
"""
from .hipapi import (AlignParams, DpResult, DpTask, HipLibraryMissing, Index, MmpParams, PairHit, RoundStats,
                     SeedStats, Context, build_library, load_library, pack_reads)

__all__ = ["AlignParams", "DpResult", "DpTask", "HipLibraryMissing", "Index", "MmpParams", "PairHit", "RoundStats",
           "SeedStats", "Context", "build_library", "load_library", "pack_reads"]
