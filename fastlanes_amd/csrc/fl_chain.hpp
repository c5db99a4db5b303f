// fl_chain.hpp -- Delta's stateful bodies and the transposes on the wave-per-block mapping (fl_widths.hpp).
//
// One kernel template, three stages, one wavefront per block, every global access 1 KiB contiguous:
//
//   SOURCE                       BODY                              SINK
//   SRC_PACKED   W packed rows   BODY_NONE                         SNK_ROWS      unpacked, transposed (FastLanes) layout
//   SRC_ROWS     unpacked block  BODY_UNDELTA  running sum + base  SNK_ORIGINAL  unpacked, ORIGINAL order (fused untranspose)
//   SRC_ORIGINAL original order  BODY_DELTA    difference to prev  SNK_PACKED    W packed rows (fused pack)
//
//   Delta::undelta_pack::<W>  (delta.rs:47-63)    = PACKED   -> UNDELTA -> ROWS
//   Delta::undelta            (delta.rs:36-45)    = ROWS     -> UNDELTA -> ROWS
//   Delta::delta              (delta.rs:24-33)    = ROWS     -> DELTA   -> ROWS
//   Transpose::untranspose    (transpose.rs:17-22)= ROWS     -> NONE    -> ORIGINAL
//   Transpose::transpose      (transpose.rs:11-15)= ORIGINAL -> NONE    -> ROWS
//   ext. undelta_pack_untranspose (delta.rs:96-100 composed)  = PACKED   -> UNDELTA -> ORIGINAL
//   ext. transpose_delta_pack     (delta.rs:88-95 composed)   = ORIGINAL -> DELTA   -> PACKED
//
// Between the stages a lane holds R = T/8 CONSECUTIVE logical rows of one cell column: lane (i = lane/8, c = lane%8)
// owns rows R*i .. R*i+R-1 of column c (16/sizeof(T) FL lanes).  The per-FL-lane chain over the T rows (row order
// matters: macros.rs:119, delta.rs:56-61) is then a local running value plus a 3-step exclusive scan of the 8 segment
// totals across lane groups (undelta), or the previous group's last row (delta).
//
// The transposes need no shuffle network either: along an FL lane's row order the original positions
// are consecutive, tau(index(r, l)) = lane_base(l) + r.  For the 32- and 64-bit types a 16-byte cell holds n = 4 / 2
// elements and the block decomposes into n x n ELEMENT TILES -- n cells (rows r..r+n-1 of lane group g) in the
// transposed layout are n cells (lanes n*g..n*g+n-1, rows r..r+n-1) in the original one -- so a lane's R rows are R/n
// tiles and the transposition is a REGISTER RENAMING (out[e].word[j] = in[j].word[e]); the original-order image lives in
// LDS with a padded line stride chosen so that the 8 lanes of a group hit 8 distinct 16-byte bank slots.  For u8 / u16 a
// cell holds more elements (16 / 8) than a lane has rows, so the tiles span lanes: there the LDS does the transposition,
// 16 single-element reads per lane per block (gather_original_cell / gather_row_cell).
// LDS is wave-local: no s_barrier.
#pragma once
#include "fl_widths.hpp"

namespace fl {

enum ChainSrc { SRC_PACKED = 0, SRC_ROWS = 1, SRC_ORIGINAL = 2 };
enum ChainBody { CHAIN_NONE = 0, CHAIN_UNDELTA = 1, CHAIN_DELTA = 2 };
enum ChainSnk { SNK_ROWS = 0, SNK_ORIGINAL = 1, SNK_PACKED = 2 };

struct ChainArgs {
    const char* in;          // packed column / unpacked column (either layout)
    char* out;               // unpacked column (either layout) / packed column
    const char* bases;       // [n_blocks][128 bytes]; unused by CHAIN_NONE
    uint64_t n_blocks;
    uint64_t tiles_per_xcd;
    unsigned window_shift;     // tile-map window (fl_kernels.hpp: xcd_tile); filled by the launcher
    unsigned width;          // SRC_PACKED / SNK_PACKED only
    // mixed-width form of the packed side (SRC_PACKED / SNK_PACKED only; fl_widths.hpp's surface): block b has widths[b] and its
    // 128*widths[b] bytes start at byte offsets[b] of the packed column.  nullptr = every block has `width`, back to back.
    const uint8_t* widths;
    const uint64_t* offsets;
    uint32_t* err_flag;      // FL_DEVERR_* of skipped blocks (may be nullptr)
    uint64_t packed_bytes;   // size of the packed column (only read when widths != nullptr)
};

// LDS image of a block in ORIGINAL order: byte a of the block lives at pad(a).  u32: +16 bytes per 128-byte line and +32 per
// KiB; u64: +16 per KiB.  With these strides the cells the 8 lanes of a group touch for one tile position (lanes n*c+e,
// c = 0..7, same rows) fall into 8 distinct 16-byte slots of the 128-byte bank window.
template <typename T> struct OriginalImage {
    static constexpr unsigned BLOCK_BYTES = WaveBlock<T>::BLOCK_BYTES;
    // u8 / u16 gather single elements (below) and keep the image linear
    __host__ __device__ static constexpr unsigned pad(unsigned a)
    {
        return sizeof(T) == 4 ? a + 16u * (a >> 7) + 32u * (a >> 10) : sizeof(T) == 8 ? a + 16u * (a >> 10) : a;
    }
    static constexpr unsigned BYTES = (pad(BLOCK_BYTES - 16u) + 16u + 255u) & ~255u;
    // byte offset (unpadded) of the original-order cell holding rows [n*q, n*q+n) of FL lane l  (transpose.rs:29-36 inverted)
    __device__ __forceinline__ static unsigned cell_of(unsigned l, unsigned q)
    {
        return (lane_base(l) * (unsigned)sizeof(T)) + 16u * q;
    }
};

// ---- u8 / u16: a cell holds more elements (16 / 8) than a lane has rows (1 / 2), so the element tiles span several lanes.
// The transposition is done by the LDS itself: 16 single-element reads per lane per block, each element fetched from where
// the OTHER layout keeps it, packed into the cell.
//   original position p of the block  <->  (FL lane l, row r):  p = lane_base(l) + r  (transpose.rs:29-36; runs of T rows)
template <typename T> __device__ __forceinline__ unsigned rows_image_byte_of_position(unsigned p)
{
    constexpr unsigned TB = sizeof(T) * 8;
    const unsigned a = p >> 6, rem = p & 63u;                 // a = l % 16
    const unsigned r = rem & (TB - 1u);                       // row inside the run
    const unsigned f = (rem - r) >> 3;                        // = FL_ORDER[l / 16]
    const unsigned l = a + 16u * ((0x73516240u >> (4u * f)) & 7u);   // FL_ORDER is its own inverse (lib.rs:53-59)
    return WaveBlock<T>::row_cell_rt(r) * 16u + l * (unsigned)sizeof(T);
}
template <typename T> __device__ __forceinline__ Cell<T> pack_elements(const uint32_t* e)
{
    Cell<T> c;
    if constexpr (sizeof(T) == 2) {
        for (int k = 0; k < 4; ++k) c.x[k] = e[2 * k] | (e[2 * k + 1] << 16);
    } else {
        for (int k = 0; k < 4; ++k) c.x[k] = e[4 * k] | (e[4 * k + 1] << 8) | (e[4 * k + 2] << 16) | (e[4 * k + 3] << 24);
    }
    return c;
}
// cell `o` (16-byte units) of the ORIGINAL order, gathered from the LDS image of the transposed rows
template <typename T> __device__ __forceinline__ Cell<T> gather_original_cell(const char* lds_rows, unsigned o)
{
    constexpr int N = 16 / (int)sizeof(T);
    uint32_t e[N];
    for (int k = 0; k < N; ++k) e[k] = *reinterpret_cast<const T*>(lds_rows + rows_image_byte_of_position<T>(N * o + k));
    return pack_elements<T>(e);
}
// cell (logical row r, lane group c) of the TRANSPOSED rows, gathered from the linear LDS image of the original order
template <typename T> __device__ __forceinline__ Cell<T> gather_row_cell(const char* lds_original, unsigned r, unsigned c)
{
    constexpr int N = 16 / (int)sizeof(T);
    uint32_t e[N];
    for (int k = 0; k < N; ++k) e[k] = *reinterpret_cast<const T*>(lds_original + (lane_base(N * c + k) + r) * (unsigned)sizeof(T));
    return pack_elements<T>(e);
}

// bytes of LDS one wavefront needs for a (source, sink) pair
template <typename T, int SRC, int SNK> constexpr unsigned chain_wave_lds()
{
    if constexpr (SRC == SRC_ORIGINAL || SNK == SNK_ORIGINAL) return OriginalImage<T>::BYTES;
    else return WaveBlock<T>::BLOCK_BYTES;
}

// value of lane (lane - 8*d) for every 32-bit word of the cell; lanes of the first d groups get zero
template <typename T> __device__ __forceinline__ Cell<T> cell_from_group_below(const Cell<T>& v, unsigned lane, unsigned d)
{
    const u32x4 w = __builtin_bit_cast(u32x4, v);
    u32x4 r;
    const int src = ((int)lane - 8 * (int)d) * 4;
    for (int k = 0; k < 4; ++k) {
        const uint32_t got = (uint32_t)__builtin_amdgcn_ds_bpermute(src < 0 ? 0 : src, (int)w[k]);
        r[k] = src < 0 ? 0u : got;
    }
    return __builtin_bit_cast(Cell<T>, r);
}

#ifdef FL_TEST_R03_REGISTER_SCAN
// KNOWN-BAD, never built into libfastlanes_amd.so (make BADSCAN=1 -> libfastlanes_amd_badscan.so only): round 3's register-only
// form of the lane-group scan (DPP + v_permlane16/32_swap instead of ds_bpermute).  It passed every per-(T, W) parity test and was
// wrong on ~3 % of the blocks of a u64 undelta_pack, differently on every run, with all CUs busy (profiles/abscan_r03.txt), and was
// dropped.  It is kept behind this macro for one purpose: to show that tests/test_gpu_full_check.py catches that class of error
// (profiles/full_check_r04.txt).  The misbehaviour depends on instruction scheduling -- after this file was cut into stages the
// same sequence stopped failing -- so the macro also plants a deterministic sparse fault (k_chain below).
template <typename T> __device__ __forceinline__ Cell<T> scan_lane_groups_r03(Cell<T> v, unsigned lane)
{
    const bool odd_row = lane & 16u, upper_half = lane & 32u;
    auto upper_group_to_both = [](const Cell<T>& x) {
        u32x4 w = __builtin_bit_cast(u32x4, x), r;
        for (int k = 0; k < 4; ++k) r[k] = (uint32_t)__builtin_amdgcn_update_dpp((int)w[k], (int)w[k], 0x108 /* row_shl:8 */, 0xF, 0xF, false);
        return r;
    };
    {
        const u32x4 w = __builtin_bit_cast(u32x4, v);
        u32x4 below;
        for (int k = 0; k < 4; ++k) below[k] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[k], 0x118 /* row_shr:8 */, 0xF, 0xF, true);
        v = v.add(__builtin_bit_cast(Cell<T>, below));
    }
    {
        const u32x4 t = upper_group_to_both(v);
        u32x4 below;
        for (int k = 0; k < 4; ++k) {
            const auto sw = __builtin_amdgcn_permlane16_swap(t[k], t[k], false, false);
            below[k] = odd_row ? (uint32_t)sw[0] : 0u;
        }
        v = v.add(__builtin_bit_cast(Cell<T>, below));
    }
    {
        const u32x4 t = upper_group_to_both(v);
        u32x4 below;
        for (int k = 0; k < 4; ++k) {
            const auto sw = __builtin_amdgcn_permlane16_swap(t[k], t[k], false, false);
            const auto hf = __builtin_amdgcn_permlane32_swap(sw[1], sw[1], false, false);
            below[k] = upper_half ? (uint32_t)hf[0] : 0u;
        }
        v = v.add(__builtin_bit_cast(Cell<T>, below));
    }
    return v;
}
#endif

// n x n element tile: in[j] = cell of row j (n lanes), out[e] = cell of lane e (n rows)  -- and back (the map is an involution)
template <typename T> __device__ __forceinline__ void tile_transpose(const Cell<T>* in, Cell<T>* out)
{
    constexpr int N = Cell<T>::NW;      // 4 dwords (u32) / 2 qwords (u64) = elements per cell for these types
    for (int e = 0; e < N; ++e)
        for (int j = 0; j < N; ++j) out[e].x[j] = in[j].x[e];
}

// Blocks per wavefront of the MIXED-WIDTH form (BPW template parameter of k_chain; every uniform-width call runs BPW = 1, the
// kernels the dispatch table was measured with).  A narrow type's block is small (u8: 1 KiB unpacked): with one block per wavefront
// the wave spends its life waiting on a single short request and the launch is millions of 4-KiB workgroups.  So a u8 / u16
// wavefront owns several CONSECUTIVE blocks and requests all of them up front by LDS-DMA, one LDS image per block, bases included
// -- the shape fl_widths.hpp's *_blocks_wave_prefetched use for the same reason (mixed-width undelta_pack u8 0.47 -> 0.58 of the
// peak, u16 0.73 -> 0.76; profiles/r04_sweep_mixed.txt).  Tried for the uniform-width calls too and NOT adopted
// (profiles/abchain_narrow_r04.txt): the per-(T,W) cell-column kernels stay far ahead for u8 either way, and u16's fused encode
// lost 8-12 % (unpacked blocks read by LDS-DMA instead of through VGPRs: the known loss of profiles/ab_ldsdma_r03.txt).
template <typename T> constexpr unsigned chain_blocks_per_wave() { return sizeof(T) == 1 ? 4u : sizeof(T) == 2 ? 2u : 1u; }

// width and packed-side byte offset of block `blk` (wave-uniform); false = the block fails a device-side precondition (flag raised)
template <typename T, int SRC, int SNK>
__device__ __forceinline__ bool chain_block_meta(const ChainArgs& a, uint64_t blk, unsigned lane, unsigned& w, uint64_t& packed_at)
{
    constexpr bool PACKED_SIDE = SRC == SRC_PACKED || SNK == SNK_PACKED;
    w = PACKED_SIDE ? a.width : (unsigned)WaveBlock<T>::TB;
    packed_at = blk * (uint64_t)(128u * w);
    if constexpr (PACKED_SIDE) {
        if (a.widths) {                                        // wave-uniform: per-block width / offset, checked on the device
            const unsigned z = opaque_zero();                  // (fl_widths.hpp: block_meta / block_precondition)
            const unsigned wv = a.widths[blk + z];
            const uint64_t ov = a.offsets[blk + z];
            w = (unsigned)__builtin_amdgcn_readfirstlane(wv);
            packed_at = wave_uniform_u64(ov);
            if (const uint32_t e = block_precondition(true, a.packed_bytes, w, packed_at, WaveBlock<T>::TB)) {
                raise_device_error(a.err_flag, e, lane);
                return false;
            }
        }
    }
    return true;
}

// descriptor over the source block
template <typename T, int SRC>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t chain_source(const ChainArgs& a, uint64_t blk, unsigned w, uint64_t packed_at)
{
    using G = WaveBlock<T>;
    const unsigned in_bytes = SRC == SRC_PACKED ? 128u * w : G::BLOCK_BYTES;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.in) + (SRC == SRC_PACKED ? packed_at : blk * (uint64_t)G::BLOCK_BYTES), 0,
                                             in_bytes, 0x00020000);
}

// One block through the three stages, cut where the wavefront's LDS traffic must be fenced so that a wavefront that owns several
// blocks (the narrow types) can take all of them through each stage before the next -- their dependent chains (LDS reads, the
// scan's ds_bpermutes) then overlap instead of queueing behind one another's fences:
//   chain_stage_source   source block -> LDS image (+ base)          [not used when the images were staged up front]
//   chain_stage_rows     LDS image -> this lane's R rows, the body applied
//   chain_stage_image    the rows -> the sink's LDS image
//   chain_stage_out      the sink image -> HBM
template <typename T, int SRC, int BODY, int RD>
__device__ __forceinline__ void chain_stage_source(const ChainArgs& a, uint64_t blk, unsigned w, uint64_t packed_at, char* lds, unsigned lane,
                                                   Cell<T>& base)
{
    using G = WaveBlock<T>;
    constexpr int TB = G::TB;
    const unsigned c16 = (lane & 7u) * 16u;
    const __amdgpu_buffer_rsrc_t in_rs = chain_source<T, SRC>(a, blk, w, packed_at);
    const unsigned w_in = SRC == SRC_PACKED ? w : (unsigned)TB;
    // LDS-DMA writes lane-linear 1 KiB pieces: usable wherever the source image is linear (not the padded original-order one)
    constexpr bool DMA = rd_is_dma(RD) && !(SRC == SRC_ORIGINAL && sizeof(T) >= 4);
    if constexpr (!DMA) {
        u32x4 img[G::GROUPS];
        static_for<G::GROUPS>([&](auto Gi) {
            constexpr int g = decltype(Gi)::value;
            if (8u * g < w_in) img[g] = __builtin_amdgcn_raw_buffer_load_b128(in_rs, lane * 16u + g * 1024u, 0, (SRC != SRC_PACKED || RD == RD_VGPR_NT) ? 2 : 0);
        });
        // base[lane] of this cell column (delta.rs:26,38,56): one 16-byte cell per column, behind the data loads
        if constexpr (BODY != CHAIN_NONE)
            base = __builtin_bit_cast(Cell<T>, *reinterpret_cast<const u32x4*>(a.bases + blk * 128u + c16 + opaque_zero()));
        static_for<G::GROUPS>([&](auto Gi) {
            constexpr int g = decltype(Gi)::value;
            if (8u * g < w_in) {
                unsigned at = lane * 16u + g * 1024u;
                if constexpr (SRC == SRC_ORIGINAL) at = OriginalImage<T>::pad(at);
                *reinterpret_cast<u32x4*>(lds + at) = img[g];
            }
        });
    } else {
        static_for<G::GROUPS>([&](auto Gi) {
            constexpr int g = decltype(Gi)::value;
            if (8u * g < w_in) dma_1k_to_lds<DMA ? RD : 0, g * 1024>(in_rs, lds, lane);
        });
        if constexpr (BODY != CHAIN_NONE)
            base = __builtin_bit_cast(Cell<T>, *reinterpret_cast<const u32x4*>(a.bases + blk * 128u + c16 + opaque_zero()));
        wait_lds_dma();
    }
}

template <typename T, int SRC, int BODY>
__device__ __forceinline__ void chain_stage_rows(unsigned w, const char* lds, unsigned lane, const Cell<T>& base, Cell<T>* x)
{
    using G = WaveBlock<T>;
    constexpr int TB = G::TB;
    constexpr int R = TB / 8;                                  // consecutive logical rows per lane
    constexpr int N = 16 / (int)sizeof(T);                     // elements per cell = tile edge
    const unsigned i = lane >> 3, c = lane & 7u, c16 = c * 16u;
    const unsigned r0 = R * i;
    // ---- this lane's R consecutive rows of cell column c --------------------------------------------------------
    if constexpr (SRC == SRC_PACKED) {
        if (w == 0) {                                          // macros.rs:118-125: every elem is 0
            static_for<R>([&](auto J) { x[decltype(J)::value] = Cell<T>::zero(); });
        } else {
            const typename G::word_t m = G::field_mask(w);
            const unsigned last = (w - 1u) * 128u;
            static_for<R>([&](auto J) {
                const unsigned bit = (r0 + decltype(J)::value) * w;
                const unsigned a0 = (bit >> G::LOG_TB) * 128u, sh = bit & (TB - 1u);
                const unsigned a1 = a0 + 128u < last ? a0 + 128u : last;
                const Cell<T> cur = __builtin_bit_cast(Cell<T>, *reinterpret_cast<const u32x4*>(lds + a0 + c16));
                const Cell<T> nxt = __builtin_bit_cast(Cell<T>, *reinterpret_cast<const u32x4*>(lds + a1 + c16));
                x[decltype(J)::value] = G::funnel(cur, nxt, sh, m);
            });
        }
    } else if constexpr (SRC == SRC_ROWS) {
        static_for<R>([&](auto J) {
            x[decltype(J)::value] = __builtin_bit_cast(Cell<T>, *reinterpret_cast<const u32x4*>(lds + G::row_cell_rt(r0 + decltype(J)::value) * 16u + c16));
        });
    } else if constexpr (sizeof(T) < 4) {
        static_for<R>([&](auto J) { x[decltype(J)::value] = gather_row_cell<T>(lds, r0 + decltype(J)::value, c); });   // transpose.rs:12-14
    } else {
        // original order: tile t of this lane = rows r0 + N*t .. + N-1 of lanes N*c .. N*c+N-1 (transpose.rs:12-14)
        static_for<R / N>([&](auto Tt) {
            constexpr int t = decltype(Tt)::value;
            Cell<T> o[N];
            static_for<N>([&](auto E) {
                const unsigned at = OriginalImage<T>::cell_of(N * c + decltype(E)::value, (r0 / N) + t);
                o[decltype(E)::value] = __builtin_bit_cast(Cell<T>, *reinterpret_cast<const u32x4*>(lds + OriginalImage<T>::pad(at)));
            });
            tile_transpose<T>(o, &x[N * t]);
        });
    }

    // ---- the body, in row order ----------------------------------------------------------------------------------
    if constexpr (BODY == CHAIN_DELTA) {
        // out[idx] = in[idx] - prev; prev = in[idx]  (delta.rs:28-30): the row before the segment is the previous lane
        // group's last row, or base for the first segment
        Cell<T> prev = cell_from_group_below<T>(x[R - 1], lane, 1);
        if (i == 0) prev = base;
        static_for<R>([&](auto J) {
            const Cell<T> cur = x[decltype(J)::value];
            x[decltype(J)::value] = cur.sub(prev);
            prev = cur;
        });
    } else if constexpr (BODY == CHAIN_UNDELTA) {
        // next = elem + prev; out[idx] = next; prev = next  (delta.rs:40-42,58-60): local running sum, then the exclusive
        // scan of the segment totals over the 8 lane groups (Hillis-Steele, 3 steps), base entering at segment 0
        if (i == 0) x[0] = x[0].add(base);
        static_for<R - 1>([&](auto J) { x[decltype(J)::value + 1] = x[decltype(J)::value + 1].add(x[decltype(J)::value]); });
#ifdef FL_TEST_R03_REGISTER_SCAN
        const Cell<T> excl = scan_lane_groups_r03<T>(x[R - 1], lane).sub(x[R - 1]);      // KNOWN-BAD test build only (see above)
#else
        Cell<T> incl = x[R - 1];
        static_for<3>([&](auto S) {
            constexpr unsigned d = 1u << decltype(S)::value;
            incl = incl.add(cell_from_group_below<T>(incl, lane, d));
        });
        const Cell<T> excl = cell_from_group_below<T>(incl, lane, 1);
#endif
        static_for<R>([&](auto J) { x[decltype(J)::value] = x[decltype(J)::value].add(excl); });
    }
}

// the rows -> the sink's LDS image (transposed rows; for u32 / u64 with SNK_ORIGINAL the padded original-order image)
template <typename T, int SNK>
__device__ __forceinline__ void chain_stage_image(const Cell<T>* x, char* lds, unsigned lane)
{
    using G = WaveBlock<T>;
    constexpr int R = G::TB / 8;
    constexpr int N = 16 / (int)sizeof(T);
    const unsigned i = lane >> 3, c = lane & 7u, c16 = c * 16u;
    const unsigned r0 = R * i;
    if constexpr (SNK == SNK_ORIGINAL && sizeof(T) >= 4) {
        static_for<R / N>([&](auto Tt) {
            constexpr int t = decltype(Tt)::value;
            Cell<T> o[N];
            tile_transpose<T>(&x[N * t], o);                   // transpose.rs:19-21
            static_for<N>([&](auto E) {
                const unsigned at = OriginalImage<T>::cell_of(N * c + decltype(E)::value, (r0 / N) + t);
                *reinterpret_cast<u32x4*>(lds + OriginalImage<T>::pad(at)) = __builtin_bit_cast(u32x4, o[decltype(E)::value]);
            });
        });
    } else {
        static_for<R>([&](auto J) {
            *reinterpret_cast<u32x4*>(lds + G::row_cell_rt(r0 + decltype(J)::value) * 16u + c16) = __builtin_bit_cast(u32x4, x[decltype(J)::value]);
        });
    }
}

template <typename T, int SNK>
__device__ __forceinline__ void chain_stage_out(const ChainArgs& a, uint64_t blk, unsigned w, uint64_t packed_at, char* lds, unsigned lane)
{
    using G = WaveBlock<T>;
    if constexpr (SNK == SNK_PACKED) {
        if (w != 0) pack_from_lds_image<T>(lds, w, a.out + packed_at, lane);   // macros.rs:52-53: W == 0 writes nothing
    } else {
        const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(a.out + blk * G::BLOCK_BYTES, 0, G::BLOCK_BYTES, 0x00020000);
        if constexpr (SNK == SNK_ORIGINAL && sizeof(T) < 4) {
            static_for<G::GROUPS>([&](auto K) {                // transpose.rs:19-21, one original-order cell per lane per KiB
                const Cell<T> v = gather_original_cell<T>(lds, lane + 64u * decltype(K)::value);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out_rs, lane * 16u + decltype(K)::value * 1024u, 0, STORE_AUX);
            });
        } else {
            static_for<G::GROUPS>([&](auto K) {
                unsigned at = lane * 16u + decltype(K)::value * 1024u;
                const unsigned to = at;
                if constexpr (SNK == SNK_ORIGINAL) at = OriginalImage<T>::pad(at);
                const u32x4 v = *reinterpret_cast<const u32x4*>(lds + at);
                __builtin_amdgcn_raw_buffer_store_b128(v, out_rs, to, 0, STORE_AUX);
            });
        }
    }
}

// One block, every stage in turn (the uniform-width kernels, the small-array batch): the wavefront's LDS image is `lds`.
template <typename T, int SRC, int BODY, int SNK, int RD>
__device__ __forceinline__ void chain_one_block(const ChainArgs& a, uint64_t blk, char* lds, unsigned lane)
{
    constexpr int R = WaveBlock<T>::TB / 8;
    // the source image is dead once every lane has taken its rows (lanes re-use other lanes' cells in the sink image, except ROWS -> ROWS)
    constexpr bool FENCE_BEFORE_IMAGE = !(SRC == SRC_ROWS && SNK == SNK_ROWS);
    unsigned w;
    uint64_t packed_at;
    if (!chain_block_meta<T, SRC, SNK>(a, blk, lane, w, packed_at)) return;
    Cell<T> base = Cell<T>::zero();
    chain_stage_source<T, SRC, BODY, RD>(a, blk, w, packed_at, lds, lane, base);
    wave_lds_fence();
    Cell<T> x[R];
    chain_stage_rows<T, SRC, BODY>(w, lds, lane, base, x);
#ifdef FL_TEST_R03_REGISTER_SCAN
    // KNOWN-BAD test build only: a deterministic SPARSE fault on top of the (scheduling-dependent) register scan -- one wrong
    // element in one of every 4 099 blocks of a u64 undelta chain.  Sampled checks miss it; the full check must not.
    if constexpr (BODY == CHAIN_UNDELTA && sizeof(T) == 8) {
        if (blk % 4099u == 4098u && lane == 13u) x[0].x[0] ^= 1u;   // first at block 4 098: beyond the small-size parity tests
    }
#endif
    if constexpr (FENCE_BEFORE_IMAGE) wave_lds_fence();
    chain_stage_image<T, SNK>(x, lds, lane);
    wave_lds_fence();
    chain_stage_out<T, SNK>(a, blk, w, packed_at, lds, lane);
}

// Several CONSECUTIVE blocks of one wavefront (the narrow types), every stage for all of them before the next: all requested up
// front by LDS-DMA (their images are linear in every layout), the bases behind them.  `lds` holds BPW images of the wavefront.
template <typename T, int SRC, int BODY, int SNK, unsigned BPW>
__device__ __forceinline__ void chain_blocks_lockstep(const ChainArgs& a, uint64_t first, unsigned count, char* lds, unsigned lane)
{
    using G = WaveBlock<T>;
    constexpr int R = G::TB / 8;
    constexpr unsigned WAVE_LDS = chain_wave_lds<T, SRC, SNK>();
    // narrow types: all of the wavefront's blocks requested up front by LDS-DMA (their images are linear in every layout),
    // the bases behind them; then every stage for all of them
    static_assert(sizeof(T) < 4, "the padded original-order image of u32 / u64 cannot be filled by LDS-DMA");
    constexpr bool FENCE_BEFORE_IMAGE = !(SRC == SRC_ROWS && SNK == SNK_ROWS);
    unsigned w[BPW];
    uint64_t packed_at[BPW];
    bool ok[BPW];
    Cell<T> base[BPW];
    // mixed widths: lane j fetches block first+j's width and offset -- one round trip for all of them -- then broadcasts
    constexpr bool PACKED_SIDE = SRC == SRC_PACKED || SNK == SNK_PACKED;
    const bool mixed = PACKED_SIDE && a.widths;            // wave-uniform
    unsigned wv = 0;
    uint64_t ov = 0;
    if (mixed) {
        const uint64_t mine = first + (lane < count ? lane : 0u);
        wv = a.widths[mine];
        ov = a.offsets[mine];
    }
    static_for<BPW>([&](auto Jt) {
        constexpr unsigned j = decltype(Jt)::value;
        ok[j] = j < count;
        w[j] = PACKED_SIDE ? a.width : (unsigned)G::TB;
        packed_at[j] = (first + j) * (uint64_t)(128u * w[j]);
        if (mixed && ok[j]) {
            w[j] = (unsigned)__builtin_amdgcn_readlane((int)wv, (int)j);
            packed_at[j] = readlane_elem<uint64_t>(ov, j);
            if (const uint32_t e = block_precondition(true, a.packed_bytes, w[j], packed_at[j], G::TB)) {
                raise_device_error(a.err_flag, e, lane);
                ok[j] = false;
            }
        }
    });
    static_for<BPW>([&](auto Jt) {
        constexpr unsigned j = decltype(Jt)::value;
        if (ok[j]) {                                        // wave-uniform
            const __amdgpu_buffer_rsrc_t in_rs = chain_source<T, SRC>(a, first + j, w[j], packed_at[j]);
            const unsigned w_in = SRC == SRC_PACKED ? w[j] : (unsigned)G::TB;
            static_for<G::GROUPS>([&](auto Gi) {
                constexpr int g = decltype(Gi)::value;
                if (8u * g < w_in) dma_1k_to_lds<RD_DMA_NT, g * 1024>(in_rs, lds + j * WAVE_LDS, lane);
            });
        }
    });
    static_for<BPW>([&](auto Jt) {
        constexpr unsigned j = decltype(Jt)::value;
        base[j] = Cell<T>::zero();
        if constexpr (BODY != CHAIN_NONE) {
            if (ok[j]) base[j] = __builtin_bit_cast(Cell<T>, *reinterpret_cast<const u32x4*>(a.bases + (first + j) * 128u + (lane & 7u) * 16u + opaque_zero()));
        }
    });
    wait_lds_dma();
    wave_lds_fence();
    Cell<T> x[BPW][R];
    static_for<BPW>([&](auto Jt) {
        constexpr unsigned j = decltype(Jt)::value;
        if (ok[j]) chain_stage_rows<T, SRC, BODY>(w[j], lds + j * WAVE_LDS, lane, base[j], x[j]);
    });
    if constexpr (FENCE_BEFORE_IMAGE) wave_lds_fence();
    static_for<BPW>([&](auto Jt) {
        constexpr unsigned j = decltype(Jt)::value;
        if (ok[j]) chain_stage_image<T, SNK>(x[j], lds + j * WAVE_LDS, lane);
    });
    wave_lds_fence();
    static_for<BPW>([&](auto Jt) {
        constexpr unsigned j = decltype(Jt)::value;
        if (ok[j]) chain_stage_out<T, SNK>(a, first + j, w[j], packed_at[j], lds + j * WAVE_LDS, lane);
    });
}

template <typename T, int SRC, int BODY, int SNK, int RD = RD_VGPR, unsigned BPW = 1>
__global__ __launch_bounds__(WG) void k_chain(ChainArgs a)
{
    using G = WaveBlock<T>;
    constexpr int R = G::TB / 8;
    constexpr unsigned WAVE_LDS = chain_wave_lds<T, SRC, SNK>();
    extern __shared__ __attribute__((aligned(16))) char lds_all[];
    constexpr unsigned TILE_BLOCKS = BPW * (WG / 64);
    const uint64_t n_tiles = (a.n_blocks + TILE_BLOCKS - 1) / TILE_BLOCKS;
    const uint64_t tile = xcd_tile(blockIdx.x, a.tiles_per_xcd, a.window_shift);
    if (tile >= n_tiles) return;
    const unsigned tid = threadIdx.x;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
    const uint64_t first = tile * TILE_BLOCKS + (uint64_t)wave * BPW;
    if (first >= a.n_blocks) return;
    char* lds = lds_all + wave * (WAVE_LDS * BPW);
    if constexpr (BPW == 1) {
        chain_one_block<T, SRC, BODY, SNK, RD>(a, first, lds, lane);
    } else {
        const uint64_t left = a.n_blocks - first;
        chain_blocks_lockstep<T, SRC, BODY, SNK, BPW>(a, first, left < BPW ? (unsigned)left : BPW, lds, lane);
    }
}

typedef hipError_t (*chain_launch_t)(const ChainArgs&, int waves, hipStream_t);

template <typename T, int SRC, int BODY, int SNK, int RD = RD_VGPR, unsigned BPW = 1>
hipError_t launch_chain(const ChainArgs& a0, int waves, hipStream_t s)
{
    if (a0.n_blocks == 0) return hipSuccess;
    ChainArgs a = a0;
    constexpr unsigned TILE_BLOCKS = BPW * (WG / 64);
    const uint64_t n_tiles = (a.n_blocks + TILE_BLOCKS - 1) / TILE_BLOCKS;
    a.tiles_per_xcd = (n_tiles + 7) / 8;
    if (a.tiles_per_xcd * 8 > 0x7fffffffull) return hipErrorInvalidValue;
    a.window_shift = tile_window_shift(SNK == SNK_PACKED ? TRAFFIC_READ : SRC == SRC_PACKED ? TRAFFIC_WRITE : TRAFFIC_BALANCED, TILE_BLOCKS);
    if constexpr (SRC == SRC_PACKED || SNK == SNK_PACKED) {
        if (a.widths) a.window_shift |= TILE_MAP_ROTATE;     // per-block widths may be periodic (fl_widths.hpp: launch_widths)
    } else {
        a.widths = nullptr;
    }
    const unsigned need = TILE_BLOCKS * chain_wave_lds<T, SRC, SNK>();
    if (waves < 3) waves = 3;
    const unsigned pad = (CU_LDS_BYTES / (unsigned)waves) & ~1023u;
    FL_LAUNCH((k_chain<T, SRC, BODY, SNK, RD, BPW>), dim3((unsigned)(a.tiles_per_xcd * 8)), dim3(WG), pad > need ? pad : need, s, a);
    return hipGetLastError();
}

// what the C ABI asks for
enum ChainOp { OP_UNDELTA_PACK = 0, OP_UNDELTA = 1, OP_DELTA = 2, OP_UNTRANSPOSE = 3, OP_TRANSPOSE = 4,
               OP_UNDELTA_PACK_UNTRANSPOSE = 5, OP_TRANSPOSE_DELTA_PACK = 6 };
template <typename T> chain_launch_t chain_launcher(int op);
// the mixed-width form (ChainArgs.widths != nullptr) of the three ops with a packed side
template <typename T> chain_launch_t chain_widths_launcher(int op);

}  // namespace fl
