// fl_dispatch.hpp -- which kernel serves a uniform-width pack / unpack / for_pack / unfor_pack call.
//
// Two kernel designs compute the same bytes (both parity-tested for every (T, W)):
//   * the per-(T,W) CELL-COLUMN kernels (fl_kernels.hpp): a thread owns a 16-byte cell column of a block, all
//     shifts are compile-time constants, a wavefront works on 8 blocks at once;
//   * the WAVE-PER-BLOCK kernels (fl_widths.hpp): one wavefront per block, runtime width, every global access
//     1 KiB contiguous, occupancy chosen at launch.
// Which one streams faster depends on (T, W, direction); the table below is the measured choice
// (tools/abuniform.hip on the same buffers, profiles/abuniform_r02*.txt): wave_policy() returns the waves/SIMD to
// run the wave-per-block kernel at, or 0 for the cell-column kernel.
#pragma once

namespace fl {

enum WaveOp { WAVE_UNPACK = 0, WAVE_PACK = 1 };

// Rows of profiles/abuniform_r02b.txt (GB/s, same buffers, every (T, W), cell-column vs wave-per-block at 3/4/5/6/8
// waves per SIMD); the wave-per-block kernel is chosen where it led by >= 2 %.  Broadly: it wins for the 32- and
// 64-bit types except at the narrowest widths (few rows per packed word: pack leaves most lanes idle) and for the
// widest 64-bit widths; wide widths like few waves in flight (3), narrow ones many (6-8).
inline int wave_policy(unsigned type_bits, unsigned w, WaveOp op)
{
    if (op == WAVE_UNPACK) {
        switch (type_bits) {
        case 64: return w <= 1 ? 3 : w == 2 ? 4 : w <= 7 ? 0 : w <= 13 ? 4 : w <= 61 ? 3 : 0;
        case 32: return w == 0 ? 3 : w <= 2 ? 8 : w <= 7 ? 0 : w <= 27 ? 4 : 3;
        case 16: return w == 0 ? 5 : w <= 5 ? 0 : w <= 10 ? 6 : 4;
        default: return w == 0 ? 6 : w <= 3 ? 0 : 8;
        }
    }
    switch (type_bits) {
    case 64: return w <= 8 ? 0 : (w % 32 == 0) ? 3 : 4;
    case 32: return w <= 2 ? 0 : w == 3 ? 8 : w <= 5 ? 6 : 4;
    case 16: return w <= 3 ? 0 : w <= 6 ? 8 : w <= 15 ? 6 : 4;
    default: return w == 8 ? 8 : 0;
    }
}

}  // namespace fl
