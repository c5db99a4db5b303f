// fl_widths.hpp -- device-resident mixed-width columns: the caller loop of the reference,
//     for b in 0..n_blocks { T::unchecked_unpack(widths[b], &packed[offsets[b]..], &mut out[b*1024..]) }
// (bitpacking.rs:109-129; pack: bitpacking.rs:76-96; loop shape benches/bitpacking.rs:80-97) with
// `widths[n_blocks]` (u8) and `offsets[n_blocks]` (u64 byte offsets) read ON THE DEVICE -- the
// surface SURVEY.md 8(b) names.  No host plan, no sort, no 32-bit window, any block count.
//
// Mapping: ONE WAVEFRONT PER 1024-VALUE BLOCK, blocks in natural column order (a workgroup takes 4
// consecutive blocks).  The width is wave-uniform (an SGPR), so the width
// dispatch of bitpacking.rs:115-128 is plain scalar arithmetic -- one kernel per element type, no
// per-width code:
//   * the packed block (128*W bytes, W rows of 8 cells) is fetched 1 KiB at a time (cell g*64+lane) into a
//     wave-private LDS image -- through VGPRs, or non-temporally by LDS-DMA where that pays (ReadPath / RD_AUTO below);
//     narrow element types keep several blocks in flight per wavefront (*_blocks_wave_prefetched);
//   * lane (i = lane/8, c = lane%8) then produces, for each 1-KiB group k of the unpacked block,
//     the cell of address-row 8k+i, column c: logical row r = row_at(8k+i) starts at bit r*W of every
//     FL lane's stream, so the lane reads packed cells (r*W/T, c) and (r*W/T + 1, c) from LDS and
//     funnel-shifts / masks all 16/sizeof(T) lanes of the cell at once (macros.rs:144-164 with
//     the shift in a register instead of a constant);
//   * every global store instruction is 1 KiB contiguous (`sc1 nt`), every load 1 KiB contiguous.
// LDS is wave-local (in-order per wave): no s_barrier.
//
// The same two kernels serve UNIFORM-width columns (widths == nullptr: every block has `uniform_width`, offsets are
// b*128*W) and the FoR bodies (refs != nullptr): for most (T, W) of the 16/32/64-bit types they out-run the
// per-(T,W) cell-column kernels of fl_kernels.hpp; the generated fl_dispatch_table.inc holds the choice (fl_dispatch.hpp).
// Per-block preconditions of the mixed-width form are checked here, on the device (block_precondition).
#pragma once
#include "fl_kernels.hpp"

namespace fl {

struct WidthsArgs {
    const char* packed;        // packed column base (unpack: in, pack: out)
    char* unpacked;            // unpacked column base
    const uint8_t* widths;     // [n_blocks]; nullptr = every block has `uniform_width`
    const uint64_t* offsets;   // [n_blocks] byte offsets into `packed`; nullptr = b * 128 * uniform_width
    uint32_t* err_flag;        // FL_DEVERR_* bits ORed in for every block that had to be skipped; may be nullptr
    const void* refs;          // FoR: references[b * ref_stride] (ffor.rs:24-50); nullptr = plain BitPacking
    uint64_t ref_stride;
    uint64_t n_blocks;
    uint64_t tiles_per_xcd;
    unsigned uniform_width;
    unsigned bpw;              // consecutive blocks per wavefront (>= 1); a workgroup takes 4*bpw blocks
    uint64_t packed_bytes;     // size of the packed column: a block must lie inside [0, packed_bytes) (only read when widths != nullptr)
    unsigned prefetch;         // bpw > 1: 1 = request all bpw blocks of the wavefront up front by LDS-DMA (one LDS image per block)
    unsigned linear_map;       // A/B tools: 1 = workgroup b takes tile b instead of the XCD-contiguous map
    unsigned window_shift;     // tile-map window (fl_kernels.hpp: xcd_tile); filled by the launcher
    unsigned nt_from;          // RD_AUTO: uniform widths >= this stream non-temporally by LDS-DMA (fl_dispatch.hpp: nt_read_from)
};

template <typename T> struct WaveBlock {
    static constexpr int TB = Elem<T>::BITS;
    static constexpr int LOG_TB = TB == 8 ? 3 : TB == 16 ? 4 : TB == 32 ? 5 : 6;
    static constexpr int GROUPS = TB / 8;                  // 1-KiB groups of an unpacked block (and of a W=T packed block)
    static constexpr unsigned BLOCK_BYTES = TB * 128;
    static constexpr int PER_S = TB / 8;
    static constexpr int KSTEP = 8 / PER_S;                // logical-row step between consecutive 1-KiB groups
    using word_t = typename Cell<T>::word_t;

    // logical row stored at address-row i (0..7) of 1-KiB group 0; group k adds k*KSTEP
    // (address-row j <-> row FL_ORDER[(j % PER_S) * KSTEP] * 8 + j / PER_S; FL_ORDER is its own inverse, lib.rs:53-59)
    __device__ __forceinline__ static unsigned row_base(unsigned i)
    {
        return ((0x73516240u >> (4 * ((i % PER_S) * KSTEP))) & 7u) * 8u + i / PER_S;
    }
    // cell index (16-byte units) of logical row r inside an unpacked block (macros.rs:20-24)
    __device__ __forceinline__ static unsigned row_cell_rt(unsigned r)
    {
        return (r & 7u) * TB + ((0x73516240u >> (4 * (r >> 3))) & 7u) * (unsigned)sizeof(T);
    }
    // element mask of `bits` bits: u64/u32 plain; u16 one 16-bit field; u8 replicated into the two 16-bit fields
    __device__ __forceinline__ static word_t field_mask(unsigned bits)
    {
        if constexpr (sizeof(T) == 8) return bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
        else if constexpr (sizeof(T) == 4) return bits >= 32 ? ~0u : ((1u << bits) - 1u);
        else if constexpr (sizeof(T) == 2) return (1u << bits) - 1u;
        else return ((1u << bits) - 1u) * 0x00010001u;
    }
    // element-replicated mask (SWAR) -- Cell<T>::rep with a runtime width
    __device__ __forceinline__ static word_t rep_mask(unsigned bits)
    {
        if constexpr (sizeof(T) == 8) return bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
        else if constexpr (sizeof(T) == 4) return bits >= 32 ? ~0u : ((1u << bits) - 1u);
        else if constexpr (sizeof(T) == 2) return ((1u << bits) - 1u) * 0x00010001u;
        else return ((1u << bits) - 1u) * 0x01010101u;
    }
    // all elements of the cell:  ((nxt:cur) >> sh) & mask   -- the straddle of macros.rs:149-161
    // (when the field does not straddle, nxt's bits land above the mask and vanish)
    __device__ __forceinline__ static Cell<T> funnel(const Cell<T>& cur, const Cell<T>& nxt, unsigned sh, word_t m)
    {
        Cell<T> r;
        if constexpr (sizeof(T) == 8) {
            for (int i = 0; i < 2; ++i) r.x[i] = ((cur.x[i] >> sh) | ((nxt.x[i] << 1) << (63u - sh))) & m;
        } else if constexpr (sizeof(T) == 4) {
            for (int i = 0; i < 4; ++i) r.x[i] = __builtin_amdgcn_alignbit(nxt.x[i], cur.x[i], sh) & m;
        } else {
            // u16 / u8, all elements of a 32-bit word at once (round 6; was: even / odd elements split with two v_perm, 8 ops per word):
            //   field = rep(mask(W)) & ((cur >> sh) | (nxt << (T - sh)))  per element
            // with 32-bit shifts the neighbouring elements' bits bleed in, but only ABOVE bit T - sh of an element (from cur >> sh) and
            // BELOW it (from nxt << (T - sh)), exactly where the other term's bits belong: `low` = the low T - sh bits of every element
            // selects between the two (v_bfi_b32), the field mask is wave-uniform -- 4 VALU per word + 3 per cell for the lane's masks
            constexpr uint32_t ALL = TB == 8 ? 0xffu : 0xffffu;
            const uint32_t rm = TB == 8 ? (m | (m << 8)) : (m | (m << 16));        // field_mask(W) -> every element (wave-uniform)
            const uint32_t b = ALL >> sh;
            const uint32_t low = __builtin_amdgcn_perm(b, b, TB == 8 ? 0x00000000u : 0x01000100u);
            const unsigned k = (unsigned)TB - sh;                                   // 1 .. T
            for (int i = 0; i < 4; ++i) {
                const uint32_t lo_part = cur.x[i] >> sh, hi_part = nxt.x[i] << k;
                r.x[i] = rm & (((lo_part ^ hi_part) & low) ^ hi_part);              // v_bfi_b32(low, lo_part, hi_part)
            }
        }
        return r;
    }
};

// A lane-uniform value the compiler cannot prove uniform: loads indexed with it go through the vector memory path
// (returned in order behind the loads issued before them) instead of a scalar load, whose `s_waitcnt lgkmcnt(0)`
// would serialise it in FRONT of the data loads (measured: unfor_pack with per-block references -6 %).
__device__ __forceinline__ unsigned opaque_zero()
{
    unsigned z = 0;
    asm volatile("" : "+v"(z));
    return z;
}

__device__ __forceinline__ uint64_t wave_uniform_u64(uint64_t v)
{
    // readfirstlane returns int: widen through uint32_t or a low word with bit 31 set sign-extends into the high one
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// lane j's value of a per-lane element (j wave-uniform), broadcast: v_readlane_b32 once or twice
template <typename T> __device__ __forceinline__ T readlane_elem(T v, unsigned j)
{
    if constexpr (sizeof(T) == 8) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)j);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), (int)j);
        return (T)(((uint64_t)hi << 32) | lo);
    } else {
        return (T)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)j);
    }
}

// Device-side error bits (include/fastlanes_amd.h: FL_DEVERR_*).  A block that violates a precondition is SKIPPED and its
// bit is ORed into *err_flag -- the device-side form of the reference's unreachable!() / debug_assert (bitpacking.rs:78-80,93,126).
constexpr uint32_t DEVERR_WIDTH = 1u, DEVERR_INDEX = 2u, DEVERR_ALIGN = 4u, DEVERR_BOUNDS = 8u;

// 0 if block `blk` of a mixed-width column may be processed, else its error bits (wave-uniform; uniform-width calls pass
// widths == nullptr and are validated on the host side of the ABI)
__device__ __forceinline__ uint32_t block_precondition(bool mixed, uint64_t packed_bytes, unsigned w, uint64_t off, unsigned type_bits)
{
    if (w > type_bits) return DEVERR_WIDTH;                                    // bitpacking.rs:93,126 unreachable!()
    if (!mixed) return 0u;
    uint32_t e = 0u;
    if (off & 15u) e |= DEVERR_ALIGN;                                           // 16-byte cells: the header's precondition
    if (off > packed_bytes || 128ull * w > packed_bytes - off) e |= DEVERR_BOUNDS;   // bitpacking.rs:78-80,111-113
    return e;
}
__device__ __forceinline__ uint32_t block_precondition(const WidthsArgs& a, unsigned w, uint64_t off, unsigned type_bits)
{
    return block_precondition(a.widths != nullptr, a.packed_bytes, w, off, type_bits);
}

__device__ __forceinline__ void raise_device_error(uint32_t* err_flag, uint32_t bits, unsigned lane)
{
    if (err_flag && lane == 0) __hip_atomic_fetch_or(err_flag, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// widths[blk] and offsets[blk] of the wavefront's block: two independent vector loads in flight together, one
// wait, then broadcast to SGPRs (wave-uniform) -- instead of a byte load, a wait, a dependent scalar load, a wait.
__device__ __forceinline__ void block_meta(const WidthsArgs& a, uint64_t blk, unsigned& w, uint64_t& off)
{
    w = a.uniform_width;
    off = blk * (uint64_t)(128u * a.uniform_width);
    if (a.widths) {
        const unsigned z = opaque_zero();
        const unsigned wv = a.widths[blk + z];
        uint64_t ov = 0;
        if (a.offsets) ov = a.offsets[blk + z];
        w = (unsigned)__builtin_amdgcn_readfirstlane(wv);
        if (a.offsets)
            off = wave_uniform_u64(ov);
        else
            off = blk * (uint64_t)(128u * w);     // offsets == nullptr with widths: only meaningful for equal widths
    } else if (a.offsets) {
        off = wave_uniform_u64(a.offsets[blk + opaque_zero()]);
    }
}

// FoR reference of the block, through the vector path (see opaque_zero); issue it AFTER the block's data loads.
template <typename T> __device__ __forceinline__ Cell<T> block_ref(const WidthsArgs& a, uint64_t blk)
{
    return Cell<T>::splat(static_cast<const T*>(a.refs)[blk * a.ref_stride + opaque_zero()]);
}

// How a wavefront's input block reaches its LDS image: through VGPRs (global load + ds_write_b128), or by LDS-DMA
// (`buffer_load_dwordx4 ... lds`: 1 KiB per instruction straight into LDS, no staging registers, no ds_write pass) with the
// default or the non-temporal cache policy.  hipcc does not order a ds_read behind a pending LDS-DMA (it emits no vmcnt
// wait for it; only the issuing wave's own vmcnt does, MI355X_MICROARCH.md item 7), hence the explicit wait.
// RD_AUTO is what the library ships (profiles/ab_ldsdma_r03.txt, two boxes, same buffers): packed input that is streamed
// once and is a sizeable share of the traffic -- every mixed-width column, and uniform widths with 2*W >= T -- gains 3-5 %
// from the NON-TEMPORAL policy on the read side (the LDS-DMA and the VGPR route tie there, the DMA route is ahead by
// 1-3 % on mixed-width columns and needs no staging registers), while narrow uniform widths and every unpacked-block read
// (pack, Delta, transposes) measured equal or worse with LDS-DMA and keep the VGPR route.
enum ReadPath { RD_AUTO = -2, RD_VGPR = -1, RD_VGPR_NT = -3, RD_DMA = 0, RD_DMA_NT = 2 };
constexpr bool rd_is_dma(int rd) { return rd >= 0; }

typedef __attribute__((address_space(3))) void* lds_dma_ptr_t;

template <int RD, int BYTE_OFF>
__device__ __forceinline__ void dma_1k_to_lds(__amdgpu_buffer_rsrc_t rs, char* lds, unsigned lane)
{
    // memory address = descriptor base + voffset + soffset + inst_offset;  LDS address = M0 base + inst_offset + lane*16:
    // the 12-bit instruction offset advances BOTH sides, so one M0 value serves 4 consecutive KiB; the 4-KiB step beyond
    // that goes into the scalar offset (memory side) and the M0 base (LDS side).  Bytes past the descriptor arrive as 0.
    constexpr int HI = BYTE_OFF & ~4095, LO = BYTE_OFF & 4095;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_dma_ptr_t)(lds + HI), 16, lane * 16u, HI, LO, RD);
}
__device__ __forceinline__ void wait_lds_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }


// the W packed rows of one block -> the wave's LDS image; FoR's reference rides behind the data loads
template <typename T, int RD>
__device__ __forceinline__ void packed_block_to_lds(const WidthsArgs& a, uint64_t blk, __amdgpu_buffer_rsrc_t rs, unsigned w,
                                                    char* lds, unsigned lane, Cell<T>& ref)
{
    using G = WaveBlock<T>;
    if constexpr (!rd_is_dma(RD)) {
        u32x4 pk[G::GROUPS];
        static_for<G::GROUPS>([&](auto Gi) {
            constexpr int g = decltype(Gi)::value;
            if (8u * g < w) pk[g] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u + g * 1024u, 0, RD == RD_VGPR_NT ? 2 : 0);
        });
        if (a.refs) ref = block_ref<T>(a, blk);                             // behind the data loads
        static_for<G::GROUPS>([&](auto Gi) {
            constexpr int g = decltype(Gi)::value;
            if (8u * g < w) *reinterpret_cast<u32x4*>(lds + lane * 16u + g * 1024u) = pk[g];
        });
    } else {
        static_for<G::GROUPS>([&](auto Gi) {
            constexpr int g = decltype(Gi)::value;
            if (8u * g < w) dma_1k_to_lds<RD, g * 1024>(rs, lds, lane);
        });
        if (a.refs) ref = block_ref<T>(a, blk);
        wait_lds_dma();
    }
}

// unchecked_unpack over per-block widths (bitpacking.rs:109-129); with a.refs also FoR::unfor_pack's body
// `out[idx] = elem + reference` (ffor.rs:46-48).  One block at a time per wavefront (a.bpw consecutive ones, 1 except
// for u8 mixed-width columns); the wave's LDS image is BLOCK_BYTES of the DYNAMIC shared memory -- the launcher pads the
// request to steer occupancy (fewer, or more, concurrent DRAM streams) without compiling per-occupancy variants.
template <typename T> __device__ __forceinline__ void unpack_zero_width_block(const WidthsArgs& a, uint64_t blk, unsigned lane);
template <typename T> __device__ __forceinline__ void unpack_lds_image_to_global(const WidthsArgs& a, uint64_t blk, unsigned w, const char* lds,
                                                                                  unsigned lane, const Cell<T>& ref);

template <typename T, int RD = RD_AUTO>
__device__ __forceinline__ void unpack_block_wave(const WidthsArgs& a, uint64_t blk, char* lds, unsigned lane)
{
    using G = WaveBlock<T>;
    constexpr int TB = G::TB;
    unsigned w;
    uint64_t off;
    block_meta(a, blk, w, off);                               // wave-uniform
    if (const uint32_t e = block_precondition(a, w, off, TB)) {   // bitpacking.rs:126 unreachable!(), :111-113
        raise_device_error(a.err_flag, e, lane);
        return;
    }
    if (w == 0) {                                             // macros.rs:118-125: every position gets 0 (+ reference)
        unpack_zero_width_block<T>(a, blk, lane);
        return;
    }
    // wave-uniform descriptor over exactly this block's 128*w bytes: cells past it read as 0, no fault
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.packed) + off, 0, 128u * w, 0x00020000);
    Cell<T> ref = Cell<T>::zero();
    if constexpr (RD == RD_AUTO) {
        if (a.widths || w >= a.nt_from) packed_block_to_lds<T, RD_DMA_NT>(a, blk, rs, w, lds, lane, ref);   // wave-uniform
        else packed_block_to_lds<T, RD_VGPR>(a, blk, rs, w, lds, lane, ref);
    } else {
        packed_block_to_lds<T, RD>(a, blk, rs, w, lds, lane, ref);
    }
    wave_lds_fence();
    unpack_lds_image_to_global<T>(a, blk, w, lds, lane, ref);
    wave_lds_fence();                                         // the image is reused by the wavefront's next block
}

// W == 0: every position gets 0 (+ FoR's reference)  (macros.rs:118-125)
template <typename T>
__device__ __forceinline__ void unpack_zero_width_block(const WidthsArgs& a, uint64_t blk, unsigned lane)
{
    using G = WaveBlock<T>;
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(
        a.unpacked + blk * G::BLOCK_BYTES, 0, G::BLOCK_BYTES, 0x00020000);
    const Cell<T> ref = a.refs ? block_ref<T>(a, blk) : Cell<T>::zero();
    static_for<G::GROUPS>([&](auto K) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ref), out_rs, lane * 16u + decltype(K)::value * 1024u, 0, STORE_AUX);
    });
}

// the LDS image of a packed block (w >= 1 rows) -> the unpacked block in HBM: lane (i, c) funnel-shifts the cell of
// address-row 8k+i, column c of every 1-KiB group k (macros.rs:144-164 with the shift in a register)
template <typename T>
__device__ __forceinline__ void unpack_lds_image_to_global(const WidthsArgs& a, uint64_t blk, unsigned w, const char* lds,
                                                           unsigned lane, const Cell<T>& ref)
{
    using G = WaveBlock<T>;
    constexpr int TB = G::TB;
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(
        a.unpacked + blk * G::BLOCK_BYTES, 0, G::BLOCK_BYTES, 0x00020000);
    const unsigned out_base = lane * 16u;
    const unsigned c16 = (lane & 7u) * 16u;
    const typename G::word_t m = G::field_mask(w);
    unsigned bit = __umul24(G::row_base(lane >> 3), w);   // both < 2^7: the full-rate 24-bit multiply
    const unsigned step = G::KSTEP * w;
    const unsigned last = (w - 1u) * 128u;
    static_for<G::GROUPS>([&](auto K) {
        const unsigned word = bit >> G::LOG_TB, sh = bit & (TB - 1u);
        const unsigned a0 = word * 128u;
        const unsigned a1 = a0 + 128u < last ? a0 + 128u : last;            // the last row never reads past the end (macros.rs:156)
        const Cell<T> cur = __builtin_bit_cast(Cell<T>, *reinterpret_cast<const u32x4*>(lds + a0 + c16));
        const Cell<T> nxt = __builtin_bit_cast(Cell<T>, *reinterpret_cast<const u32x4*>(lds + a1 + c16));
        Cell<T> v = G::funnel(cur, nxt, sh, m);
        if (a.refs) v = v.add(ref);                                         // ffor.rs:46-48
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out_rs, out_base + decltype(K)::value * 1024u, 0, STORE_AUX);
        bit += step;
    });
}

// Several consecutive blocks per wavefront, ALL their packed rows requested up front: one LDS image per block, filled by
// LDS-DMA (no staging registers, so the depth costs only LDS), one wait, then the blocks are decoded back to back.  A narrow
// element type's block is small (u8: at most 1 KiB packed, 1 KiB unpacked): with one block in flight per wavefront the wave
// spends its life waiting for a single short request; with `count` of them in flight it has count times the bytes
// outstanding for the same number of resident waves.
template <typename T>
__device__ __forceinline__ void unpack_blocks_wave_prefetched(const WidthsArgs& a, uint64_t first, unsigned count, char* lds, unsigned lane)
{
    using G = WaveBlock<T>;
    constexpr int TB = G::TB;
    // per-lane metadata of block first + lane (count <= 16 <= 64 lanes); broadcast per block with readlane
    const uint64_t mine = first + (lane < count ? lane : 0u);
    unsigned wv = a.uniform_width;
    if (a.widths) wv = a.widths[mine];
    uint64_t ov = mine * (uint64_t)(128u * wv);
    if (a.offsets) ov = a.offsets[mine];
    T rv = 0;                                                 // FoR: lane j holds block first+j's reference, fetched with the metadata
    if (a.refs) rv = static_cast<const T*>(a.refs)[mine * a.ref_stride];
    for (unsigned j = 0; j < count; ++j) {                    // wave-uniform loop
        const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)wv, (int)j);
        const uint64_t off = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ov >> 32), (int)j) << 32) |
                             (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ov, (int)j);
        if (block_precondition(a, w, off, TB) || w == 0) continue;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.packed) + off, 0, 128u * w, 0x00020000);
        char* img = lds + j * G::BLOCK_BYTES;
        static_for<G::GROUPS>([&](auto Gi) {
            constexpr int g = decltype(Gi)::value;
            if (8u * g < w) dma_1k_to_lds<RD_DMA_NT, g * 1024>(rs, img, lane);
        });
    }
    wait_lds_dma();
    wave_lds_fence();
    for (unsigned j = 0; j < count; ++j) {
        const uint64_t blk = first + j;
        const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)wv, (int)j);
        const uint64_t off = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ov >> 32), (int)j) << 32) |
                             (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ov, (int)j);
        if (const uint32_t e = block_precondition(a, w, off, TB)) {   // bitpacking.rs:126 unreachable!(), :111-113
            raise_device_error(a.err_flag, e, lane);
            continue;
        }
        if (w == 0) {
            unpack_zero_width_block<T>(a, blk, lane);
            continue;
        }
        const Cell<T> ref = Cell<T>::splat(readlane_elem<T>(rv, j));
        unpack_lds_image_to_global<T>(a, blk, w, lds + j * G::BLOCK_BYTES, lane, ref);
    }
}

// The same with the block count known at compile time and every block of the wavefront known to be valid (round 6).  A narrow type's
// wavefront lives on instructions as much as on memory: per 6 KiB of traffic the u8 kernel above issued 223 VALU + 334 SALU + 79 branches
// where u32's one-block wavefront issues 145 + 182 + 31, every instruction costs a wavefront ~7 cycles of its life, and with all eight
// wave slots taken nothing else hides it (profiles/r06_wavelife_mixed.txt).  Here the preconditions of all BPW blocks are ONE per-lane
// test and a ballot (any failure -> `false`: the caller runs the general function, which skips and reports block by block), the loops
// are unrolled with the blocks' widths / offsets in SGPRs across the wait, one store descriptor serves the wavefront's BPW consecutive
// output blocks, W = 0 needs no path of its own (an empty descriptor fetches nothing and the field mask is 0), and FoR's `+ reference`
// is a template parameter instead of a branch per cell.
template <typename T, unsigned BPW, bool REFS>
__device__ __forceinline__ void unpack_images_to_global(const WidthsArgs& a, uint64_t first, const unsigned (&w)[BPW], const char* lds, unsigned lane, T rv)
{
    using G = WaveBlock<T>;
    constexpr int TB = G::TB;
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(a.unpacked + first * G::BLOCK_BYTES, 0, BPW * G::BLOCK_BYTES, 0x00020000);
    const unsigned out_base = lane * 16u, c16 = (lane & 7u) * 16u, rb = G::row_base(lane >> 3);
    static_for<(int)BPW>([&](auto J) {
        constexpr unsigned j = decltype(J)::value;
        const unsigned wj = w[j];
        const char* img = lds + j * G::BLOCK_BYTES;
        const typename G::word_t m = G::field_mask(wj);
        unsigned bit = __umul24(rb, wj);
        const unsigned step = G::KSTEP * wj, last = (wj - 1u) * 128u;         // W = 0: `last` wraps, the reads stay inside the image, m = 0
        Cell<T> ref = Cell<T>::zero();
        if constexpr (REFS) ref = Cell<T>::splat(readlane_elem<T>(rv, j));
        static_for<G::GROUPS>([&](auto K) {
            const unsigned word = bit >> G::LOG_TB, sh = bit & (TB - 1u);
            const unsigned a0 = word * 128u;
            const unsigned a1 = a0 + 128u < last ? a0 + 128u : last;        // the last row never reads past the end (macros.rs:156)
            const Cell<T> cur = __builtin_bit_cast(Cell<T>, *reinterpret_cast<const u32x4*>(img + a0 + c16));
            const Cell<T> nxt = __builtin_bit_cast(Cell<T>, *reinterpret_cast<const u32x4*>(img + a1 + c16));
            Cell<T> v = G::funnel(cur, nxt, sh, m);
            if constexpr (REFS) v = v.add(ref);                             // ffor.rs:46-48
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out_rs, out_base + (j * G::BLOCK_BYTES + decltype(K)::value * 1024u), 0,
                                                   STORE_AUX);
            bit += step;
        });
    });
}

template <typename T, unsigned BPW>
__device__ __forceinline__ bool unpack_blocks_wave_static(const WidthsArgs& a, uint64_t first, char* lds, unsigned lane)
{
    using G = WaveBlock<T>;
    constexpr int TB = G::TB;
    const bool owner = lane < BPW;
    const uint64_t mine = first + (owner ? lane : 0u);
    unsigned wv = a.uniform_width;
    if (a.widths) wv = a.widths[mine];
    uint64_t ov = mine * (uint64_t)(128u * wv);
    if (a.offsets) ov = a.offsets[mine];
    T rv = 0;                                                 // FoR: lane j holds block first+j's reference, fetched with the metadata
    if (a.refs) rv = static_cast<const T*>(a.refs)[mine * a.ref_stride];
    // bitpacking.rs:126 unreachable!(), :111-113 -- all BPW blocks at once, lane j judging block first + j
    if (__builtin_amdgcn_ballot_w64(block_precondition(a, wv, ov, TB) != 0u)) return false;
    unsigned w[BPW];
    static_for<(int)BPW>([&](auto J) {
        constexpr unsigned j = decltype(J)::value;
        w[j] = (unsigned)__builtin_amdgcn_readlane((int)wv, (int)j);
        const uint64_t off = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ov >> 32), (int)j) << 32) |
                             (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ov, (int)j);
        // wave-uniform descriptor over exactly this block's 128*w bytes: cells past it read as 0, no fault; W = 0 fetches nothing
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.packed) + off, 0, 128u * w[j], 0x00020000);
        char* img = lds + j * G::BLOCK_BYTES;
        static_for<G::GROUPS>([&](auto Gi) {
            constexpr int g = decltype(Gi)::value;
            if (8u * g < w[j]) dma_1k_to_lds<RD_DMA_NT, g * 1024>(rs, img, lane);
        });
    });
    wait_lds_dma();
    wave_lds_fence();
    if (a.refs) unpack_images_to_global<T, BPW, true>(a, first, w, lds, lane, rv);      // wave-uniform
    else unpack_images_to_global<T, BPW, false>(a, first, w, lds, lane, rv);
    return true;
}

// Wavefront -> blocks: workgroup `tile` (XCD-contiguous map) owns 4*bpw consecutive blocks, wavefront `wave` the bpw
// consecutive ones starting at tile*4*bpw + wave*bpw.
template <typename T, typename F>
__device__ __forceinline__ void for_each_block_of_wave(const WidthsArgs& a, F&& f)
{
    using G = WaveBlock<T>;
    extern __shared__ __attribute__((aligned(16))) char lds_all[];
    const unsigned tile_blocks = a.bpw * (WG / 64);
    const uint64_t n_tiles = (a.n_blocks + tile_blocks - 1) / tile_blocks;
    // XCD-contiguous map (workgroup ids go round-robin over the 8 XCDs: XCD x owns one contiguous eighth of the column);
    // linear_map (A/B tools only) = workgroup b takes tile b
    const uint64_t tile = a.linear_map ? (uint64_t)blockIdx.x : xcd_tile(blockIdx.x, a.tiles_per_xcd, a.window_shift);
    if (tile >= n_tiles) return;
    const unsigned tid = threadIdx.x;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
    char* lds = lds_all + wave * G::BLOCK_BYTES * (a.prefetch ? a.bpw : 1u);
    const uint64_t first = tile * tile_blocks + (uint64_t)wave * a.bpw;
    if (first >= a.n_blocks) return;
    const uint64_t left = a.n_blocks - first;
    f(first, left < a.bpw ? (unsigned)left : a.bpw, lds, lane);
}

template <typename T, int RD = RD_AUTO>
__global__ __launch_bounds__(WG) void k_unpack_widths(WidthsArgs a)
{
    for_each_block_of_wave<T>(a, [&](uint64_t first, unsigned count, char* lds, unsigned lane) {
        if (a.prefetch && count > 1) {
            if constexpr (sizeof(T) <= 2) {                  // the shipped shapes of the narrow types, every block valid: the unrolled form
                // (u32 with two blocks per wavefront through the same code ties with one block at 6 waves: 0.848 / 0.848, round 6)
                if (count == a.bpw) {
                    if constexpr (sizeof(T) == 1) {
                        if (a.bpw == 4 && unpack_blocks_wave_static<T, 4>(a, first, lds, lane)) return;
                    }
                    if (a.bpw == 2 && unpack_blocks_wave_static<T, 2>(a, first, lds, lane)) return;
                }
            }
            unpack_blocks_wave_prefetched<T>(a, first, count, lds, lane);
            return;
        }
        for (unsigned j = 0; j < count; ++j) unpack_block_wave<T, RD>(a, first + j, lds, lane);
    });
}

// The W packed rows of one block assembled from the wave's LDS image of the UNPACKED block (transposed layout, cell of
// logical row r at row_cell(r)): lane (i, c) builds packed cells (w = i + 8m, c) -- word w of an FL lane's stream holds bits
// [w*T, (w+1)*T), i.e. the fields of rows floor(w*T/W) .. floor(((w+1)*T-1)/W) (macros.rs:72-92 regrouped by destination
// word instead of by source row) -- and stores them 1 KiB-contiguously.  Requires 1 <= w <= T.
//
// Narrow widths (w < 8) would leave most lanes idle that way (w = 1: one packed row, 8 of 64 lanes, 32 rows each).  They
// take the other route: lane (i, c) owns the R = T/8 CONSECUTIVE rows R*i .. R*i+R-1 of column c, whose fields are one
// contiguous chunk of R*w <= T bits of every FL lane's stream, starting at bit R*i*w: the lane splices its chunk together,
// and the chunks of the lanes that share a packed word are merged with LDS atomic ORs (ds_or_b32) into a zeroed image of the
// packed block (re-using the first KiB of the unpacked image, which is dead once every lane holds its rows).
template <typename T, bool REFS = false>
__device__ __forceinline__ void pack_narrow_from_lds_image(char* lds, unsigned w, char* packed_block, unsigned lane, const Cell<T>& ref = Cell<T>::zero())
{
    using G = WaveBlock<T>;
    using word_t = typename G::word_t;
    constexpr int TB = G::TB;
    constexpr int R = TB / 8;
    constexpr int NW = Cell<T>::NW;
    const unsigned c16 = (lane & 7u) * 16u, i = lane >> 3;
    // chunk = field(row R*i) | field(row R*i+1) << w | ...   (macros.rs:73,79 for R consecutive rows)
    const word_t mw = G::rep_mask(w);
    word_t chunk[NW];
    for (int x = 0; x < NW; ++x) chunk[x] = 0;
    static_for<R>([&](auto J) {
        Cell<T> s = __builtin_bit_cast(Cell<T>, *reinterpret_cast<const u32x4*>(lds + G::row_cell_rt(R * i + decltype(J)::value) * 16u + c16));
        if constexpr (REFS) s = s.sub(ref);                   // ffor.rs:32-34 (before the mask of macros.rs:73)
        for (int x = 0; x < NW; ++x) chunk[x] |= (s.x[x] & mw) << (decltype(J)::value * w);
    });
    wave_lds_fence();                                         // every lane holds its rows: the image may be overwritten
    *reinterpret_cast<u32x4*>(lds + lane * 16u) = u32x4{0, 0, 0, 0};
    wave_lds_fence();
    // the chunk starts at stream bit R*i*w = word (i*w)/8, bit ((i*w)%8)*R, and may run into the next word (macros.rs:84-92)
    const unsigned wl = (i * w) >> 3, sb = ((i * w) & 7u) * R;
    const unsigned room = TB - sb;                            // bits of word wl at and above sb
    const unsigned len = R * w;
    uint32_t* lo_at = reinterpret_cast<uint32_t*>(lds + wl * 128u + c16);
    const word_t mlo = G::rep_mask(len < room ? len : room);
    for (int x = 0; x < NW; ++x) {
        const word_t v = (chunk[x] & mlo) << sb;
        if constexpr (sizeof(word_t) == 8) {
            __hip_atomic_fetch_or(lo_at + 2 * x, (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_or(lo_at + 2 * x + 1, (uint32_t)(v >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        } else {
            __hip_atomic_fetch_or(lo_at + x, (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    }
    if (len > room) {                                         // carry into word wl + 1 (sb > 0 here)
        const word_t mhi = G::rep_mask(len - room);
        uint32_t* hi_at = lo_at + 32;                         // next packed row: +128 bytes
        for (int x = 0; x < NW; ++x) {
            const word_t v = (chunk[x] >> room) & mhi;
            if constexpr (sizeof(word_t) == 8) {
                __hip_atomic_fetch_or(hi_at + 2 * x, (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_fetch_or(hi_at + 2 * x + 1, (uint32_t)(v >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            } else {
                __hip_atomic_fetch_or(hi_at + x, (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
    }
    wave_lds_fence();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(packed_block, 0, 128u * w, 0x00020000);
    const u32x4 out = *reinterpret_cast<const u32x4*>(lds + lane * 16u);
    __builtin_amdgcn_raw_buffer_store_b128(out, rs, lane * 16u, 0, STORE_AUX);   // cells past 128*w bytes are dropped by the descriptor
}

// ceil(2^20 / w): floor(n / w) = (n * RECIP20[w]) >> 20 for every n < 4160 and w >= 8 (the error term n * (RECIP20[w] * w - 2^20) / (2^20 * w)
// stays below 1/252 < 1/w) -- a lane's row index without an integer division (two ~12-instruction sequences per packed word otherwise)
__device__ constexpr uint32_t RECIP20[65] = {
    0,      1048576, 524288, 349526, 262144, 209716, 174763, 149797, 131072, 116509, 104858, 95326, 87382, 80660, 74899, 69906, 65536,
    61681,  58255,   55189,  52429,  49933,  47663,  45591,  43691,  41944,  40330,  38837, 37450, 36158, 34953, 33826, 32768,
    31776,  30841,   29960,  29128,  28340,  27595,  26887,  26215,  25576,  24967,  24386, 23832, 23302, 22796, 22311, 21846,
    21400,  20972,   20561,  20165,  19785,  19419,  19066,  18725,  18397,  18079,  17773, 17477, 17190, 16913, 16645, 16384};

// REFS: FoR's `input[idx] - reference` (ffor.rs:32-34) applied to every cell as it is read from the image (a cell is read by the one or two
// lanes whose packed words hold bits of its row) -- for images that arrived by LDS-DMA and were never in registers
template <typename T, bool REFS = false>
__device__ __forceinline__ void pack_from_lds_image(char* lds, unsigned w, char* packed_block, unsigned lane, const Cell<T>& ref = Cell<T>::zero())
{
    using G = WaveBlock<T>;
    using word_t = typename G::word_t;
    constexpr int TB = G::TB;
    if (w < 8u) {                                             // wave-uniform
        pack_narrow_from_lds_image<T, REFS>(lds, w, packed_block, lane, ref);
        return;
    }
    const unsigned c16 = (lane & 7u) * 16u, i = lane >> 3;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(packed_block, 0, 128u * w, 0x00020000);
    const uint32_t recip = RECIP20[w];                        // wave-uniform
    for (unsigned m8 = 0; m8 < w; m8 += 8) {                  // wave-uniform trip count: ceil(w/8) groups of 8 packed rows
        const unsigned wd = m8 + i;                           // this lane's packed word-row
        Cell<T> acc = Cell<T>::zero();
        if (wd < w) {
            const unsigned lo_bit = wd * TB;
            unsigned r = (lo_bit * recip) >> 20;              // = lo_bit / w: first row with bits in this word
            const unsigned r_end = ((lo_bit + TB - 1u) * recip) >> 20;   // last such row (inclusive)
            for (; r <= r_end; ++r) {
                Cell<T> s = __builtin_bit_cast(Cell<T>, *reinterpret_cast<const u32x4*>(lds + G::row_cell_rt(r) * 16u + c16));
                if constexpr (REFS) s = s.sub(ref);
                const unsigned fb = r * w;                    // first stream bit of row r's field
                if (fb >= lo_bit) {                           // field starts in this word: (src & mask(keep)) << shift  (macros.rs:73,79)
                    const unsigned sh = fb - lo_bit;
                    const unsigned keep = w < TB - sh ? w : TB - sh;
                    const word_t mk = G::rep_mask(keep);
                    for (int x = 0; x < Cell<T>::NW; ++x) acc.x[x] |= (s.x[x] & mk) << sh;
                } else {                                      // carry of a straddling field: (src & mask(W)) >> (W - rem)  (macros.rs:92)
                    const unsigned sh = lo_bit - fb;
                    const word_t mk = G::rep_mask(w - sh);
                    for (int x = 0; x < Cell<T>::NW; ++x) acc.x[x] |= (s.x[x] >> sh) & mk;
                }
            }
        }
        // rows past the block's 128*w bytes fall outside the descriptor and are dropped
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc), rs, wd * 128u + c16, 0, STORE_AUX);
    }
}

// unchecked_pack over per-block widths (bitpacking.rs:76-96); with a.refs also FoR::for_pack's body
// `input[idx] - reference` (ffor.rs:32-34).  The unpacked block is parked in the wave's LDS image; lane (i, c)
// then assembles packed cells (w = i + 8m, c): word w of an FL lane's stream holds bits [w*T, (w+1)*T), i.e. the
// fields of rows floor(w*T/W) .. floor(((w+1)*T-1)/W) (macros.rs:72-92 regrouped by destination word instead of by
// source row).
template <typename T, int RD = RD_VGPR>
__device__ __forceinline__ void pack_block_wave(const WidthsArgs& a, uint64_t blk, char* lds, unsigned lane)
{
    using G = WaveBlock<T>;
    constexpr int TB = G::TB;
    unsigned w;
    uint64_t off;
    block_meta(a, blk, w, off);                               // wave-uniform
    if (const uint32_t e = block_precondition(a, w, off, TB)) {   // bitpacking.rs:93 unreachable!(), :78-80
        raise_device_error(a.err_flag, e, lane);
        return;
    }
    if (w == 0) return;                                       // macros.rs:52-53: W == 0 writes nothing
    const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc(
        a.unpacked + blk * G::BLOCK_BYTES, 0, G::BLOCK_BYTES, 0x00020000);
    if (!rd_is_dma(RD) || a.refs) {                           // FoR subtracts on the way into the image: through VGPRs
        u32x4 un[G::GROUPS];
        static_for<G::GROUPS>([&](auto K) {
            un[decltype(K)::value] = __builtin_amdgcn_raw_buffer_load_b128(in_rs, decltype(K)::value * 1024u + lane * 16u, 0, 2 /* nt */);
        });
        const Cell<T> ref = a.refs ? block_ref<T>(a, blk) : Cell<T>::zero();   // behind the data loads
        static_for<G::GROUPS>([&](auto K) {
            Cell<T> v = __builtin_bit_cast(Cell<T>, un[decltype(K)::value]);
            if (a.refs) v = v.sub(ref);                                         // ffor.rs:32-34 (before the mask of macros.rs:73)
            *reinterpret_cast<u32x4*>(lds + lane * 16u + decltype(K)::value * 1024u) = __builtin_bit_cast(u32x4, v);
        });
    } else {
        static_for<G::GROUPS>([&](auto K) { dma_1k_to_lds<rd_is_dma(RD) ? RD : 0, decltype(K)::value * 1024>(in_rs, lds, lane); });
        wait_lds_dma();
    }
    wave_lds_fence();
    pack_from_lds_image<T>(lds, w, const_cast<char*>(a.packed) + off, lane);
    wave_lds_fence();                                         // the image is reused by the wavefront's next block
}

// pack's counterpart of unpack_blocks_wave_prefetched: the `count` unpacked blocks of the wavefront arrive by LDS-DMA, one
// image each, before the first one is packed.  FoR's `input[idx] - reference` (ffor.rs:32-34) is applied by the packer as it reads
// the image's cells (round 6; an in-place pass over the image -- one more LDS read, write and fence per block -- cost for_pack_widths u8
// 6 % against pack_widths).
template <typename T>
__device__ __forceinline__ void pack_blocks_wave_prefetched(const WidthsArgs& a, uint64_t first, unsigned count, char* lds, unsigned lane)
{
    using G = WaveBlock<T>;
    constexpr int TB = G::TB;
    const uint64_t mine = first + (lane < count ? lane : 0u);
    unsigned wv = a.uniform_width;
    if (a.widths) wv = a.widths[mine];
    uint64_t ov = mine * (uint64_t)(128u * wv);
    if (a.offsets) ov = a.offsets[mine];
    T rv = 0;                                                 // FoR: lane j holds block first+j's reference
    if (a.refs) rv = static_cast<const T*>(a.refs)[mine * a.ref_stride];
    for (unsigned j = 0; j < count; ++j) {                    // wave-uniform loop
        const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc(
            a.unpacked + (first + j) * G::BLOCK_BYTES, 0, G::BLOCK_BYTES, 0x00020000);
        char* img = lds + j * G::BLOCK_BYTES;
        static_for<G::GROUPS>([&](auto K) { dma_1k_to_lds<RD_DMA_NT, decltype(K)::value * 1024>(in_rs, img, lane); });
    }
    wait_lds_dma();
    wave_lds_fence();
    for (unsigned j = 0; j < count; ++j) {
        const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)wv, (int)j);
        const uint64_t off = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ov >> 32), (int)j) << 32) |
                             (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ov, (int)j);
        if (const uint32_t e = block_precondition(a, w, off, TB)) {   // bitpacking.rs:93 unreachable!(), :78-80
            raise_device_error(a.err_flag, e, lane);
            continue;
        }
        if (w == 0) continue;                                 // macros.rs:52-53: W == 0 writes nothing
        if (a.refs) pack_from_lds_image<T, true>(lds + j * G::BLOCK_BYTES, w, const_cast<char*>(a.packed) + off, lane, Cell<T>::splat(readlane_elem<T>(rv, j)));
        else pack_from_lds_image<T>(lds + j * G::BLOCK_BYTES, w, const_cast<char*>(a.packed) + off, lane);
    }
}

template <typename T, int RD = RD_VGPR>
__global__ __launch_bounds__(WG) void k_pack_widths(WidthsArgs a)
{
    for_each_block_of_wave<T>(a, [&](uint64_t first, unsigned count, char* lds, unsigned lane) {
        if (a.prefetch && count > 1) {
            pack_blocks_wave_prefetched<T>(a, first, count, lds, lane);
            return;
        }
        for (unsigned j = 0; j < count; ++j) pack_block_wave<T, RD>(a, first + j, lds, lane);
    });
}

typedef hipError_t (*widths_launch_t)(const WidthsArgs&, int waves, hipStream_t);

// Occupancy is steered at launch: the kernels use BLOCK_BYTES of dynamic LDS per wavefront, and the launcher
// pads the request so that exactly `waves` wavefronts per SIMD (= workgroups per CU) fit the CU's 160 KiB.
// Mixed-width columns (profiles/abmixed_r02a.txt, abmixed_r02b.txt, abmixed_r02d.txt; u32, 9 765 625 blocks, widths
// 1 + b mod 32 and seeded-random): one block per wavefront at 5-6 waves/SIMD; more blocks per wavefront with a register
// prefetch of the next block measured 1-5 % slower, the round-1 bucketed plan kernel 5-10 % slower; a bare 33:64
// read:write stream of the same bytes reaches 6.23-6.53 TB/s where this kernel reaches 6.37-6.66.  The shipped
// occupancy / blocks-per-wavefront per type live in fl_dispatch.hpp (mixed_waves, mixed_blocks_per_wave).
constexpr unsigned CU_LDS_BYTES = 160 * 1024;
template <typename T> inline unsigned widths_lds_bytes(int waves, unsigned images_per_wave = 1)
{
    const unsigned need = (WG / 64) * WaveBlock<T>::BLOCK_BYTES * images_per_wave;
    if (waves < 3) waves = 3;                     // 53 KiB per workgroup: stays below the 64 KiB default dynamic-LDS limit
    unsigned pad = (CU_LDS_BYTES * (unsigned)WG / ((unsigned)waves * 256u)) & ~1023u;
    return pad > need ? pad : need;
}

template <typename T, bool PACK, int RD = (PACK ? RD_VGPR : RD_AUTO)>
hipError_t launch_widths(const WidthsArgs& a0, int waves, hipStream_t s)
{
    if (a0.n_blocks == 0) return hipSuccess;
    WidthsArgs a = a0;
    if (a.bpw == 0) a.bpw = 1;
    const uint64_t tile_blocks = (uint64_t)a.bpw * (WG / 64);
    const uint64_t n_tiles = (a.n_blocks + tile_blocks - 1) / tile_blocks;
    a.tiles_per_xcd = (n_tiles + 7) / 8;
    if (a.tiles_per_xcd * 8 > 0x7fffffffull) return hipErrorInvalidValue;   // > 2^33 blocks in one launch
    a.window_shift = tile_window_shift(PACK ? WIN_PACK : WIN_UNPACK, WaveBlock<T>::TB, (unsigned)tile_blocks);
    if (a.widths) a.window_shift |= TILE_MAP_ROTATE;         // per-block widths may be periodic: keep CUs from locking onto one phase
    const dim3 grid((unsigned)(a.tiles_per_xcd * 8));
    if (a.bpw < 2 || a.bpw > 16) a.prefetch = 0;
    const unsigned lds = widths_lds_bytes<T>(waves, a.prefetch ? a.bpw : 1u);
    if (lds > 64 * 1024) return hipErrorInvalidValue;         // beyond the default dynamic-LDS limit
    if constexpr (PACK) FL_LAUNCH((k_pack_widths<T, RD>), grid, dim3(WG), lds, s, a);
    else FL_LAUNCH((k_unpack_widths<T, RD>), grid, dim3(WG), lds, s, a);
    return hipGetLastError();
}

template <typename T> widths_launch_t widths_launcher(bool pack);

}  // namespace fl
