"""fastlanes_amd -- MI355X-native FastLanes codec hot path (pack / unpack /
FoR / Delta / transpose of 1024-value blocks) behind the reference's trait
names.  The compute lives in libfastlanes_amd.so (hand-written gfx950 HIP
kernels, C ABI in include/fastlanes_amd.h); this package is the thin host-side
mirror of the reference interface.  No CPU fallback exists."""
from ._lib import LIB_PATH, exported_symbols, load  # noqa: F401
from .codec import (Batch, BitPacking, Delta, FastLanesError, FoR, MixedWidthPlan,  # noqa: F401
                    Transpose, for_pack_widths, for_widths, pack_widths, packed_len, transpose_delta_pack_widths,
                    undelta_pack_widths, unfor_pack_widths, unpack_single_widths, unpack_widths, widths_to_offsets)

FL_ORDER = (0, 4, 2, 6, 1, 5, 3, 7)  # lib.rs:22

__all__ = ["BitPacking", "FoR", "Delta", "Transpose", "FastLanesError", "packed_len", "MixedWidthPlan", "Batch",
           "unpack_widths", "pack_widths", "unpack_single_widths", "widths_to_offsets",
           "unfor_pack_widths", "for_pack_widths", "for_widths", "undelta_pack_widths", "transpose_delta_pack_widths",
           "FL_ORDER", "load", "exported_symbols", "LIB_PATH"]
