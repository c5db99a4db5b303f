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

enum WaveOp { WAVE_UNPACK = 0, WAVE_PACK = 1, WAVE_UNDELTA_PACK = 2, WAVE_UNDELTA = 3, WAVE_DELTA = 4,
              WAVE_UNTRANSPOSE = 5, WAVE_TRANSPOSE = 6, WAVE_UNDELTA_PACK_UNTRANSPOSE = 7, WAVE_TRANSPOSE_DELTA_PACK = 8 };

// profiles/abuniform_r02b.txt and abuniform_r02c.txt are two boxes' full sweeps (GB/s on the same buffers, every
// (T, W), cell-column vs wave-per-block at 3/4/5/6/8 waves per SIMD; r02c packs full-entropy values).  The
// wave-per-block kernel is chosen where it led on BOTH boxes (typically by 3-10 %); where the boxes disagree (e.g.
// u32 unpack at W >= 21: +3 % on one, -2 % on the other) the cell-column kernel stays.  Broadly: wave-per-block wins
// for the 32- and 64-bit types except at the narrowest widths of pack (few rows per packed word leave most lanes
// idle) and the widest widths; wide widths like few waves in flight (3-4), narrow ones many (6-8).
// Delta's stateful bodies (fl_chain.hpp) -- profiles/abchain_r02*.txt
// The chain kernels do more per block (a second pass through LDS and a cross-lane scan), so they only pay where
// the cell-column kernels are furthest from the memory system's ceiling: delta / undelta of the 16-bit type (+6-7 %),
// of u32 / u64 at 3 waves/SIMD (+1-4 %), u8 delta (+5 %), and undelta_pack of u64 at mid widths (+3 %).
inline int chain_policy(unsigned type_bits, unsigned w, WaveOp op)
{
    if (op >= WAVE_UNTRANSPOSE) {
        // original-order forms (profiles/abchain_r02c.txt, abchain_r02d.txt, abchain_r02e.txt): transpose / untranspose
        // +5...10 % for every type (u64 / u32 at 3 waves/SIMD, u16 at 4, u8 at 8); fused decode to original order
        // +4...13 % from mid widths up for u32 / u64; the narrow types' fused decode stays on the cell-column kernels
        if (op == WAVE_TRANSPOSE || op == WAVE_UNTRANSPOSE) return type_bits >= 32 ? 3 : type_bits == 16 ? 4 : 8;
        if (op == WAVE_UNDELTA_PACK_UNTRANSPOSE) {
            if (type_bits < 32) return 0;
            return type_bits == 32 ? (w < 10 ? 0 : w <= 16 ? 6 : 4) : (w < 12 ? 0 : w < 32 ? 4 : 3);
        }
        // fused encode (with the atomic-OR narrow pack: u32 W=3 +11 %, u64 W=60 +18 %, mid widths +2...4 %)
        if (w == 0) return 0;
        return type_bits == 64 ? 4 : type_bits == 32 ? (w <= 16 ? 6 : 4) : type_bits == 16 ? (w >= 12 ? 6 : 0) : 0;
    }
    if (op == WAVE_UNDELTA_PACK) return (type_bits == 64 && w >= 12 && w <= 48) ? (w >= 32 ? 3 : 4) : 0;
    switch (type_bits) {
    case 64: return 3;
    case 32: return 3;
    case 16: return 4;
    default: return op == WAVE_DELTA ? 8 : 0;
    }
}

inline int wave_policy(unsigned type_bits, unsigned w, WaveOp op)
{
    if (op >= WAVE_UNDELTA_PACK) return chain_policy(type_bits, w, op);
    if (op == WAVE_UNPACK) {
        switch (type_bits) {
        case 64: return w <= 1 ? 3 : w <= 17 ? 4 : w <= 48 ? 3 : 0;   // W = 15..17: 3 and 4 waves trade places with the column size; 4 at 10 M blocks
        case 32: return w == 0 ? 3 : w <= 4 ? 8 : w <= 7 ? 0 : w <= 20 ? 4 : 0;
        case 16: return w == 0 ? 5 : w <= 5 ? 0 : w <= 10 ? 6 : 4;
        default: return w == 0 ? 6 : w <= 3 ? 0 : 8;
        }
    }
    switch (type_bits) {
    case 64: return w <= 7 ? 0 : (w % 32 == 0) ? 3 : 4;
    case 32: return w == 0 ? 0 : w <= 2 ? 6 : 4;      // narrow widths: chunks merged with LDS atomic ORs (profiles/abuniform_r02e.txt)
    case 16: return w <= 3 ? 0 : w <= 15 ? 6 : 4;
    default: return w >= 7 ? 8 : 0;
    }
}

// MIXED-width columns (profiles/abmixed_r02e.txt: every type, seeded-random widths 1..T, blocks per wavefront x waves/SIMD).
// A u8 block is only 1 KiB -- too little work to amortise a wavefront's start-up -- so a u8 wavefront walks 8 consecutive
// blocks (unpack 4 625 -> 5 673 GB/s, pack 4 342 -> 5 461); for the wider types one block per wavefront stays best
// (u16 6 171, u32 6 193, u64 6 556 GB/s).  The narrow types want every wave slot (8), u32 / u64 unpack 6.
inline unsigned mixed_blocks_per_wave(unsigned type_bits) { return type_bits == 8 ? 8u : 1u; }
inline int mixed_waves(unsigned type_bits, bool pack) { return (pack || type_bits <= 16) ? 8 : 6; }

}  // namespace fl
