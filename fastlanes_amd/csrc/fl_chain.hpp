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
#include <atomic>

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
    unsigned nt_from;        // RD_AUTO: uniform widths >= this stream non-temporally by LDS-DMA (fl_dispatch.hpp: nt_read_from)
};

// LDS image of a block in ORIGINAL order: byte a of the block lives at pad(a).  u32: +16 bytes per 128-byte line and +32 per
// KiB; u64: +16 per KiB.  With these strides the cells the 8 lanes of a group touch for one tile position (lanes n*c+e,
// c = 0..7, same rows) fall into 8 distinct 16-byte slots of the 128-byte bank window.
template <typename T> struct OriginalImage {
    static constexpr unsigned BLOCK_BYTES = WaveBlock<T>::BLOCK_BYTES;
    // u8 / u16 gather elements (below).  u16: +16 bytes per KiB -- a lane group's gather reads dword 32k + 4 FL_ORDER[c/2] + i of KiB
    // c%2 (k = element, c = cell column, i = lane group): the two KiB would share every bank (2-way conflict on each of the reads,
    // 38 % of the LDS time of transpose_delta_pack u16 in round 5: profiles/r05_sq_mixed_final.txt); 4 dwords apart they interleave.
    // u8 keeps the image linear (its lanes read bytes of shared dwords).
    __host__ __device__ static constexpr unsigned pad(unsigned a)
    {
        return sizeof(T) == 4 ? a + 16u * (a >> 7) + 32u * (a >> 10) : sizeof(T) == 8 ? a + 16u * (a >> 10) : sizeof(T) == 2 ? a + 16u * (a >> 10) : a;
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
// (The rows image a u16 block is GATHERED from, SNK_ORIGINAL, has the same 2-way conflict -- lanes o and o + 1 read rows k and k + 8,
// 128 bytes apart -- but padding it (+16 bytes per 128-byte row) LOST 3 .. 9 % on undelta_pack_untranspose u16, uniform and mixed widths:
// profiles/r06_ab_u16_chain.txt; the padded ds_write_b128 of the rows costs more than the gathers' conflicts.  It stays linear.)
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
// cell (logical row r, lane group c) of the TRANSPOSED rows, gathered from the LDS image of the original order
template <typename T> __device__ __forceinline__ Cell<T> gather_row_cell(const char* lds_original, unsigned r, unsigned c)
{
    constexpr int N = 16 / (int)sizeof(T);
    uint32_t e[N];
    for (int k = 0; k < N; ++k) e[k] = *reinterpret_cast<const T*>(lds_original + OriginalImage<T>::pad((lane_base(N * c + k) + r) * (unsigned)sizeof(T)));
    return pack_elements<T>(e);
}
// u16: a lane's TWO rows r0 (even), r0 + 1 of cell column c at once -- along an FL lane the rows are consecutive in the original
// order (tau(index(r, l)) = lane_base(l) + r), so one 32-bit read per FL lane fetches both rows' elements: 8 reads instead of 16,
// the halves sorted into the two cells by v_perm
__device__ __forceinline__ void gather_row_pair_cells(const char* lds_original, unsigned r0, unsigned c, Cell<uint16_t>& lo, Cell<uint16_t>& hi)
{
    uint32_t e[8];
    for (int k = 0; k < 8; ++k)
        e[k] = *reinterpret_cast<const uint32_t*>(lds_original + OriginalImage<uint16_t>::pad((lane_base(8u * c + k) + r0) * 2u));
    for (int m = 0; m < 4; ++m) {
        lo.x[m] = __builtin_amdgcn_perm(e[2 * m + 1], e[2 * m], 0x05040100u);     // low halves:  e[2m].lo | e[2m+1].lo << 16
        hi.x[m] = __builtin_amdgcn_perm(e[2 * m + 1], e[2 * m], 0x07060302u);     // high halves: e[2m].hi | e[2m+1].hi << 16
    }
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
    if constexpr (RD == RD_AUTO) {
        // what fl_widths.hpp's RD_AUTO does for plain unpack (profiles/ab_ldsdma_r03.txt): packed input that is a sizeable share of the
        // traffic (2 W >= T, or a mixed-width column) streams NON-TEMPORALLY by LDS-DMA, narrow widths through VGPRs.  Round 6: the
        // Delta decoders read their packed rows with the default policy at every width until now -- 3-6 % behind unpack of the same
        // column at wide widths at every occupancy, in both kernel designs (profiles/r06_undelta_occupancy.txt).
        if (SRC == SRC_PACKED && (a.widths || w >= a.nt_from)) chain_stage_source<T, SRC, BODY, RD_DMA_NT>(a, blk, w, packed_at, lds, lane, base);
        else chain_stage_source<T, SRC, BODY, RD_VGPR>(a, blk, w, packed_at, lds, lane, base);
        return;
    }
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
            // (the instruction offset advances the memory and the LDS side alike: a padded image shifts the LDS base instead)
            constexpr unsigned shift = SRC == SRC_ORIGINAL ? OriginalImage<T>::pad(g * 1024u) - g * 1024u : 0u;
            if (8u * g < w_in) dma_1k_to_lds<DMA ? RD : 0, g * 1024>(in_rs, lds + shift, lane);
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
    } else if constexpr (sizeof(T) == 2) {
        gather_row_pair_cells(lds, r0, c, x[0], x[1]);                                                                  // transpose.rs:12-14
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
        Cell<T> incl = x[R - 1];
        static_for<3>([&](auto S) {
            constexpr unsigned d = 1u << decltype(S)::value;
            incl = incl.add(cell_from_group_below<T>(incl, lane, d));
        });
        const Cell<T> excl = cell_from_group_below<T>(incl, lane, 1);
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

// the sink image of an UNPACKED block -> HBM through `out_rs` (a descriptor over the block; an empty one drops the stores)
// `voff` = this lane's byte offset of the block's first KiB inside the descriptor (lane * 16 for a descriptor over the block itself)
template <typename T, int SNK>
__device__ __forceinline__ void chain_stage_out_unpacked(__amdgpu_buffer_rsrc_t out_rs, const char* lds, unsigned lane, unsigned voff)
{
    using G = WaveBlock<T>;
    if constexpr (SNK == SNK_ORIGINAL && sizeof(T) < 4) {
        static_for<G::GROUPS>([&](auto K) {                // transpose.rs:19-21, one original-order cell per lane per KiB
            const Cell<T> v = gather_original_cell<T>(lds, lane + 64u * decltype(K)::value);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out_rs, voff + decltype(K)::value * 1024u, 0, STORE_AUX);
        });
    } else {
        static_for<G::GROUPS>([&](auto K) {
            unsigned at = lane * 16u + decltype(K)::value * 1024u;
            if constexpr (SNK == SNK_ORIGINAL) at = OriginalImage<T>::pad(at);
            const u32x4 v = *reinterpret_cast<const u32x4*>(lds + at);
            __builtin_amdgcn_raw_buffer_store_b128(v, out_rs, voff + decltype(K)::value * 1024u, 0, STORE_AUX);
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
        chain_stage_out_unpacked<T, SNK>(out_rs, lds, lane, lane * 16u);
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
    static_assert(sizeof(T) < 4 || (SRC != SRC_ORIGINAL && SNK != SNK_ORIGINAL), "the padded original-order image of u32 / u64 cannot be filled by LDS-DMA");
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
                constexpr unsigned shift = SRC == SRC_ORIGINAL ? OriginalImage<T>::pad(g * 1024u) - g * 1024u : 0u;
                if (8u * g < w_in) dma_1k_to_lds<RD_DMA_NT, g * 1024>(in_rs, lds + j * WAVE_LDS + shift, lane);
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
        const unsigned count = left < BPW ? (unsigned)left : BPW;
        chain_blocks_lockstep<T, SRC, BODY, SNK, BPW>(a, first, count, lds, lane);
    }
}

typedef hipError_t (*chain_launch_t)(const ChainArgs&, int waves, hipStream_t);

// the row of the generated window table (fl_dispatch.hpp) a (source, body, sink) triple belongs to
template <int SRC, int BODY, int SNK> constexpr WindowOp chain_window_op()
{
    if (SNK == SNK_PACKED) return WIN_TRANSPOSE_DELTA_PACK;
    if (SRC == SRC_PACKED) return SNK == SNK_ORIGINAL ? WIN_UNDELTA_PACK_UNTRANSPOSE : WIN_UNDELTA_PACK;
    if (BODY == CHAIN_UNDELTA) return WIN_UNDELTA;
    if (BODY == CHAIN_DELTA) return WIN_DELTA;
    return SNK == SNK_ORIGINAL ? WIN_UNTRANSPOSE : WIN_TRANSPOSE;
}

template <typename T, int SRC, int BODY, int SNK, int RD = RD_VGPR, unsigned BPW = 1>
hipError_t launch_chain(const ChainArgs& a0, int waves, hipStream_t s)
{
    if (a0.n_blocks == 0) return hipSuccess;
    ChainArgs a = a0;
    constexpr unsigned TILE_BLOCKS = BPW * (WG / 64);
    const uint64_t n_tiles = (a.n_blocks + TILE_BLOCKS - 1) / TILE_BLOCKS;
    a.tiles_per_xcd = (n_tiles + 7) / 8;
    if (a.tiles_per_xcd * 8 > 0x7fffffffull) return hipErrorInvalidValue;
    a.window_shift = tile_window_shift(chain_window_op<SRC, BODY, SNK>(), WaveBlock<T>::TB, TILE_BLOCKS);
    if constexpr (SRC == SRC_PACKED || SNK == SNK_PACKED) {
        if (a.widths) a.window_shift |= TILE_MAP_ROTATE;     // per-block widths may be periodic (fl_widths.hpp: launch_widths)
    } else {
        a.widths = nullptr;
    }
    const unsigned need = TILE_BLOCKS * chain_wave_lds<T, SRC, SNK>();
    if (waves < 3) waves = 3;
    const unsigned pad = (CU_LDS_BYTES * (unsigned)WG / ((unsigned)waves * 256u)) & ~1023u;
    FL_LAUNCH((k_chain<T, SRC, BODY, SNK, RD, BPW>), dim3((unsigned)(a.tiles_per_xcd * 8)), dim3(WG), pad > need ? pad : need, s, a);
    return hipGetLastError();
}

// ---- COLUMN LANES (round 5): Delta's decode over a mixed-width u8 column, EIGHT consecutive blocks per wavefront --------------------
// The lockstep form above gives a lane R = T/8 rows of a block -- for u8 ONE row -- so the running sum of delta.rs:56-61 runs ACROSS
// lanes: base into group 0, a 3-step Hillis-Steele scan of the 8 lane groups by ds_bpermute (16 of them per block) with a SWAR byte
// add (6 VALU per word) at every step: ~200 VALU instructions per 1-KiB block, VALU issue 0.75-0.81 of nominal, 0.56-0.60 of the HBM
// peak (profiles/r05_sq_mixed_lockstep.txt).
// Here lane (j = lane/8, c = lane%8) owns cell column c of block first+j for ALL T rows -- the ownership of the per-(T,W) cell-column
// kernels (fl_device.hpp), but with the block's width in a VGPR: the lane funnel-shifts its T fields out of its block's LDS image
// (macros.rs:144-164 with per-lane shift and mask) and the chain is a thread-local running value: no cross-lane traffic, no scan
// (58 VALU per block).  The decoded rows go back into the same image IN PLACE (a lane only ever touches its own 16-byte column of
// its own block), and every block leaves 1 KiB-contiguously as before.
// The running value is kept SPLIT -- even bytes and odd bytes in the low bytes of two u16x2 registers, exactly what the funnel's two
// v_perm produce anyway -- so an add is two v_pk_add_u16 (carries run into the unused high bytes) instead of the 6-op SWAR form,
// and one v_perm per word merges them for the store.
// That alone bought nothing (profiles/exp_columns_r05.txt, step 1): with a tile = a fresh workgroup the kernel is bound by the
// wavefront's serial life, not by instructions.  The kernel below is therefore PIPELINED (k_chain_columns_pipelined).
// Tried for u16 as well (16 rows per lane, 162 VGPRs) and not adopted: u16's lockstep kernel is not instruction-bound and stays ahead.
template <typename T> struct ColumnSum;        // running value of one cell column, += one row's fields, -> the row's cell
template <> struct ColumnSum<uint8_t> {
    uint32_t ev[4], od[4];
    __device__ __forceinline__ void start(const Cell<uint8_t>& base)
    {
        for (int k = 0; k < 4; ++k) { ev[k] = base.x[k]; od[k] = base.x[k] >> 8; }   // the bytes above each low byte are never read
    }
    // fields of the row from (nxt:cur) >> sh, masked with m (WaveBlock<u8>::field_mask); returns the row's cell
    __device__ __forceinline__ Cell<uint8_t> step(const Cell<uint8_t>& cur, const Cell<uint8_t>& nxt, unsigned sh, uint32_t m)
    {
        Cell<uint8_t> r;
        for (int k = 0; k < 4; ++k) {
            const uint32_t e = (__builtin_amdgcn_perm(nxt.x[k], cur.x[k], 0x06020400u) >> sh) & m;   // nxt.b2:cur.b2 | nxt.b0:cur.b0
            const uint32_t o = (__builtin_amdgcn_perm(nxt.x[k], cur.x[k], 0x07030501u) >> sh) & m;   // nxt.b3:cur.b3 | nxt.b1:cur.b1
            ev[k] = __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, ev[k]) + __builtin_bit_cast(u16x2, e)));
            od[k] = __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, od[k]) + __builtin_bit_cast(u16x2, o)));
            r.x[k] = __builtin_amdgcn_perm(od[k], ev[k], 0x06020400u);                                // od.b2 : ev.b2 : od.b0 : ev.b0
        }
        return r;
    }
};
constexpr unsigned COLUMN_LANES_BPW = 8;       // 64 lanes = 8 blocks x 8 cell columns

// width and packed-side byte offset of block first+j in lane j (j < count), as loaded -- possibly still in flight
struct ColumnMeta { unsigned wv; uint64_t ov; };
__device__ __forceinline__ ColumnMeta columns_meta_load(const ChainArgs& a, uint64_t first, unsigned count, unsigned lane)
{
    ColumnMeta m{a.width, 0};
    if (a.widths && count) {                                // wave-uniform
        const uint64_t mine = first + (lane < count ? lane : 0u);
        m.wv = a.widths[mine];
        m.ov = a.offsets[mine];
    }
    return m;
}

// ---- the PIPELINED form of the column-lanes kernel (round 5) -----------------------------------------------------------------
// One wavefront per workgroup, PERSISTENT (the grid is what is resident at once; workgroup b walks tile-map slots b, b + grid, ..: the
// grid is a multiple of 8, so it stays on its XCD and inside that XCD's run of the column), software-pipelined over its tiles of 8
// blocks:   metadata of tile k+2 | packed rows + bases of tile k+1 in flight (into VGPRs: 8 blocks x 16 bytes per lane per KiB)
//           | tile k decoded out of LDS and stored.
// A wavefront therefore has a tile's worth of reads outstanding ALL the time instead of only while it has nothing else to do: with a
// tile = a fresh workgroup (k_chain_columns above, and the lockstep kernel) a wavefront's life is metadata round trip, data round
// trip, decode, stores, strictly one after the other, and the LDS that bounds how many blocks a CU holds is idle for the first half.
// Everything is an ordinary tracked load (no LDS-DMA): hipcc places the partial s_waitcnt vmcnt(N) of the loop-carried loads itself.
// The loads of a block that is absent or fails a precondition go through an EMPTY descriptor (no traffic, zeros back; its stores
// likewise), so the loop body is straight-line code.
// ---- u8's transposition in REGISTERS --------------------------------------------------------------------------------------------
// Along FL lane l's row order the original positions are consecutive (tau(index(r, l)) = lane_base(l) + r, transpose.rs:29-36): for
// u8 the 8 rows of FL lane l = 16c + e are the 8 CONSECUTIVE BYTES at  e * 64 + FL_ORDER[c] * 8  of the original-order block.  A lane
// that owns cell column c (FL lanes 16c .. 16c+15) for all 8 rows therefore needs sixteen 8-byte reads (ds_read_b64) instead of 128
// single-byte ones, and a 16 x 8 byte transpose in registers: eight 4 x 4 byte tiles, 8 v_perm_b32 each.
// The original-order image of block j sits at j * ORIGINAL_U8_STRIDE: 64 bytes of padding per image, so that the 8 lane groups of a
// wavefront (8 blocks, every group reading one 64-byte span per instruction) spread over the banks instead of piling onto 16 of them.
constexpr unsigned ORIGINAL_U8_STRIDE = 1024 + 64;

// 4 x 4 byte transpose (an involution): a[i] = 4 bytes of thing i  ->  o[j] = byte j of things 0..3
__device__ __forceinline__ void transpose4x4_bytes(const uint32_t* a, uint32_t* o)
{
    const uint32_t t0 = __builtin_amdgcn_perm(a[1], a[0], 0x05010400u);      // a0.b0 a1.b0 a0.b1 a1.b1
    const uint32_t t1 = __builtin_amdgcn_perm(a[1], a[0], 0x07030602u);      // a0.b2 a1.b2 a0.b3 a1.b3
    const uint32_t u0 = __builtin_amdgcn_perm(a[3], a[2], 0x05010400u);
    const uint32_t u1 = __builtin_amdgcn_perm(a[3], a[2], 0x07030602u);
    o[0] = __builtin_amdgcn_perm(u0, t0, 0x05040100u);                       // a0.b0 a1.b0 a2.b0 a3.b0
    o[1] = __builtin_amdgcn_perm(u0, t0, 0x07060302u);
    o[2] = __builtin_amdgcn_perm(u1, t1, 0x05040100u);
    o[3] = __builtin_amdgcn_perm(u1, t1, 0x07060302u);
}
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
// runs[e] = the 8 rows of FL lane 16c+e (8 bytes)  <->  rows[r] = the cell of row r (16 FL lanes); the map is its own inverse
__device__ __forceinline__ void runs_to_rows_u8(const u32x2_t* runs, Cell<uint8_t>* rows)
{
    for (int k = 0; k < 4; ++k)
        for (int h = 0; h < 2; ++h) {
            uint32_t a[4], o[4];
            for (int i = 0; i < 4; ++i) a[i] = runs[4 * k + i][h];
            transpose4x4_bytes(a, o);
            for (int j = 0; j < 4; ++j) rows[4 * h + j].x[k] = o[j];
        }
}
__device__ __forceinline__ void rows_to_runs_u8(const Cell<uint8_t>* rows, u32x2_t* runs)
{
    for (int k = 0; k < 4; ++k)
        for (int h = 0; h < 2; ++h) {
            uint32_t a[4], o[4];
            for (int j = 0; j < 4; ++j) a[j] = rows[4 * h + j].x[k];
            transpose4x4_bytes(a, o);
            for (int i = 0; i < 4; ++i) runs[4 * k + i][h] = o[i];
        }
}
// byte offset of this lane's first run inside an original-order image (lane = 8j + c): FL_ORDER[c] * 8; run e is 64 * e further
__device__ __forceinline__ unsigned first_run_of_column(unsigned c) { return ((0x73516240u >> (4u * c)) & 7u) * 8u; }

template <typename T> struct ColumnTile {
    u32x4 p[COLUMN_LANES_BPW][WaveBlock<T>::GROUPS];        // block j, packed KiB g: this lane's 16 bytes
    Cell<T> base;                                           // base[lane] of this lane's cell column (delta.rs:56)
    unsigned wm;                                            // this lane's block's width (0 if the block is not decoded)
    unsigned okmask;                                        // wave-uniform: bit j = block first+j is decoded
    unsigned count;                                         // blocks of the tile that exist
    uint64_t first;
};

// Requests tile [first, first + count): lane j checks block first+j's width and offset ITSELF (the per-block preconditions of
// fl_widths.hpp: block_precondition, bitpacking.rs:126 / :111-113, as vector code: no scalar branches), a ballot makes the tile's
// ok-mask, and the per-block descriptors are built from readlane'd values that are already zeroed for a block that is not decoded.
template <typename T>
__device__ __forceinline__ void columns_issue(const ChainArgs& a, uint64_t first, unsigned count, const ColumnMeta& meta, unsigned lane,
                                              ColumnTile<T>& d, uint32_t& lane_errors)
{
    using G = WaveBlock<T>;
    constexpr unsigned BPW = COLUMN_LANES_BPW;
    const bool mixed = a.widths != nullptr;                 // wave-uniform
    const unsigned jm = lane >> 3;
    unsigned wv = a.width;
    uint64_t ov = (first + lane) * (uint64_t)(128u * a.width);
    uint32_t e = 0;
    if (mixed) {
        wv = meta.wv;
        ov = meta.ov;
        const uint32_t misaligned = (ov & 15u) ? DEVERR_ALIGN : 0u;
        const uint32_t outside = (ov > a.packed_bytes || 128ull * wv > a.packed_bytes - ov) ? DEVERR_BOUNDS : 0u;
        e = wv > (unsigned)G::TB ? DEVERR_WIDTH : (misaligned | outside);
    }
    const bool mine = lane < count;
    e = mine ? e : 0u;
    lane_errors |= e;
    const bool okl = mine && e == 0;
    wv = okl ? wv : 0u;
    ov = okl ? ov : 0ull;
    d.first = first;
    d.count = count;
    d.okmask = (unsigned)__builtin_amdgcn_ballot_w64(okl);  // lanes >= 8 are never `mine` (count <= 8)
    static_for<BPW>([&](auto Jt) {
        constexpr unsigned j = decltype(Jt)::value;
        const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)wv, (int)j);
        const uint64_t packed_at = readlane_elem<uint64_t>(ov, j);
        // an absent / refused block: empty descriptor -- no traffic, zeros back
        const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.in) + packed_at, 0, 128u * w, 0x00020000);
        static_for<G::GROUPS>([&](auto Gi) {
            constexpr int g = decltype(Gi)::value;
            d.p[j][g] = __builtin_amdgcn_raw_buffer_load_b128(in_rs, lane * 16u + g * 1024u, 0, 2 /* nt */);
        });
    });
    d.wm = (unsigned)__builtin_amdgcn_ds_bpermute((int)(jm * 4u), (int)wv);
    const bool okm = (d.okmask >> jm) & 1u;
    // the bases of the wavefront's 8 blocks are 1 KiB contiguous; lanes of absent blocks ask for nothing
    const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.bases) + first * 128u, 0, count * 128u, 0x00020000);
    d.base = __builtin_bit_cast(Cell<T>, __builtin_amdgcn_raw_buffer_load_b128(b_rs, okm ? lane * 16u : 0xfffffff0u, 0, 0));
}

template <typename T, int SNK>
__device__ __forceinline__ void columns_consume(const ChainArgs& a, const ColumnTile<T>& d, char* lds, unsigned lane)
{
    using G = WaveBlock<T>;
    constexpr int TB = G::TB;
    constexpr unsigned BPW = COLUMN_LANES_BPW, IMG = G::BLOCK_BYTES;
    const unsigned jm = lane >> 3, c16 = (lane & 7u) * 16u;
    static_for<BPW>([&](auto Jt) {
        static_for<G::GROUPS>([&](auto Gi) {
            *reinterpret_cast<u32x4*>(lds + decltype(Jt)::value * IMG + decltype(Gi)::value * 1024u + lane * 16u) = d.p[decltype(Jt)::value][decltype(Gi)::value];
        });
    });
    wave_lds_fence();
    char* img = lds + jm * IMG + c16;
    const unsigned wm = d.wm;
    const typename G::word_t m = G::field_mask(wm);         // 0 for W == 0: every elem is 0 (macros.rs:118-125)
    const unsigned last = (wm ? wm - 1u : 0u) * 128u;
    Cell<T> x[TB];
    ColumnSum<T> sum;
    sum.start(d.base);
    static_for<TB>([&](auto R) {
        constexpr unsigned r = decltype(R)::value;
        const unsigned bit = r * wm;
        const unsigned a0 = (bit >> G::LOG_TB) * 128u, sh = bit & (TB - 1u);
        const unsigned a1 = a0 + 128u < last ? a0 + 128u : last;               // the last row never reads past the end (macros.rs:156)
        const Cell<T> cur = __builtin_bit_cast(Cell<T>, *reinterpret_cast<const u32x4*>(img + a0));
        const Cell<T> nxt = __builtin_bit_cast(Cell<T>, *reinterpret_cast<const u32x4*>(img + a1));
        x[r] = sum.step(cur, nxt, sh, m);
    });
    // ONE descriptor over the tile's 8 consecutive unpacked blocks; a block that is not decoded keeps its bytes: its stores go out of range
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(a.out + d.first * G::BLOCK_BYTES, 0, d.count * G::BLOCK_BYTES, 0x00020000);
    if constexpr (SNK == SNK_ORIGINAL && sizeof(T) == 1) {
        // original order (transpose.rs:19-21) by a register transpose: this lane's 16 FL lanes' runs of 8 rows, 8 bytes each, into a
        // (padded) original-order image -- the packed images are dead once every lane holds its rows
        u32x2_t runs[16];
        rows_to_runs_u8(x, runs);
        wave_lds_fence();
        char* at = lds + jm * ORIGINAL_U8_STRIDE + first_run_of_column(lane & 7u);
        static_for<16>([&](auto E) { *reinterpret_cast<u32x2_t*>(at + 64 * decltype(E)::value) = runs[decltype(E)::value]; });
        wave_lds_fence();
        static_for<BPW>([&](auto Jt) {
            constexpr unsigned j = decltype(Jt)::value;
            const bool ok = (d.okmask >> j) & 1u;           // wave-uniform
            const u32x4 v = *reinterpret_cast<const u32x4*>(lds + j * ORIGINAL_U8_STRIDE + lane * 16u);
            __builtin_amdgcn_raw_buffer_store_b128(v, out_rs, ok ? lane * 16u + j * G::BLOCK_BYTES : 0xfffff000u, 0, STORE_AUX);
        });
    } else {
        // in place: a lane only ever touches its own column of its own block (rows of an absent block are never stored)
        static_for<TB>([&](auto R) {
            constexpr unsigned r = decltype(R)::value;
            *reinterpret_cast<u32x4*>(img + Elem<T>::row_cell(r) * 16) = __builtin_bit_cast(u32x4, x[r]);
        });
        wave_lds_fence();
        static_for<BPW>([&](auto Jt) {
            constexpr unsigned j = decltype(Jt)::value;
            const bool ok = (d.okmask >> j) & 1u;           // wave-uniform
            chain_stage_out_unpacked<T, SNK>(out_rs, lds + j * IMG, lane, ok ? lane * 16u + j * G::BLOCK_BYTES : 0xfffff000u);
        });
    }
    wave_lds_fence();
}

template <typename T, int SNK>
__global__ __launch_bounds__(64) void k_chain_columns_pipelined(ChainArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr unsigned BPW = COLUMN_LANES_BPW;
    const unsigned tiles_per_xcd = (unsigned)a.tiles_per_xcd;                 // the launcher keeps 8 * tiles_per_xcd below 2^31
    const unsigned slots = tiles_per_xcd * 8, grid = gridDim.x;
    const unsigned n_tiles = (unsigned)((a.n_blocks + BPW - 1) / BPW);
    const unsigned lane = threadIdx.x;
    auto blocks_of = [&](unsigned t, uint64_t& first, unsigned& count) {      // slot t of the tile map; count = 0: nothing there
        const unsigned tile = t < slots ? xcd_tile<uint32_t>(t, tiles_per_xcd, a.window_shift) : n_tiles;
        const bool there = tile < n_tiles;
        first = there ? (uint64_t)tile * BPW : 0;
        const uint64_t left = a.n_blocks - first;
        count = there ? (left < BPW ? (unsigned)left : BPW) : 0u;
    };
    uint32_t lane_errors = 0;
    unsigned t1 = blockIdx.x + grid;                        // slot of the tile after the current one
    uint64_t first0, first1;
    unsigned count0, count1;
    blocks_of(blockIdx.x, first0, count0);
    blocks_of(t1, first1, count1);
    const ColumnMeta m0 = columns_meta_load(a, first0, count0, lane);
    ColumnMeta m1 = columns_meta_load(a, first1, count1, lane);
    ColumnTile<T> cur;
    columns_issue<T>(a, first0, count0, m0, lane, cur, lane_errors);
    for (;;) {
        const unsigned t2 = t1 + grid;
        uint64_t first2;
        unsigned count2;
        blocks_of(t2, first2, count2);
        const ColumnMeta m2 = columns_meta_load(a, first2, count2, lane);      // tile k+2's metadata
        ColumnTile<T> nxt;
        columns_issue<T>(a, first1, count1, m1, lane, nxt, lane_errors);       // tile k+1's packed rows and bases
        columns_consume<T, SNK>(a, cur, lds, lane);                            // tile k
        if (t1 >= slots) break;
        cur = nxt;
        m1 = m2;
        first1 = first2;
        count1 = count2;
        t1 = t2;                                            // t1 < slots < 2^31 and grid < 2^31: no wrap-around
    }
    if (__builtin_amdgcn_ballot_w64(lane_errors != 0)) {
        for (int o = 32; o; o >>= 1) lane_errors |= (uint32_t)__shfl_xor((int)lane_errors, o);
        raise_device_error(a.err_flag, lane_errors, lane);
    }
}

// Workgroups of a persistent single-wavefront kernel that are resident at once on the current device: CUs x min(wanted per CU, what
// fits).  The two device queries cost microseconds and never change for a kernel on a device: cached per kernel (one slot per
// template instantiation) and device id.
template <auto KERNEL>
inline hipError_t persistent_grid(unsigned lds, int per_cu, unsigned& grid)
{
    // one slot per (kernel, device id): threads that drive different devices (examples/multi_gpu_decode.c) do not evict each other's
    // entry, and a reader can never pair one device's figures with another's id.  packed = 1 + cus + (fit << 16); 0 = not queried yet
    constexpr int MAX_DEVICES = 16;
    static std::atomic<uint64_t> slots[MAX_DEVICES];         // one array per instantiation = per kernel (KERNEL is a non-type parameter)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool cached = dev >= 0 && dev < MAX_DEVICES;
    uint64_t packed = cached ? slots[dev].load(std::memory_order_relaxed) : 0;
    int cus = (int)((packed - 1) & 0xffffu), fit = (int)((packed - 1) >> 16);
    if (packed == 0) {
        e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, KERNEL, 64, lds);
        if (e != hipSuccess) return e;
        if (cached && cus > 0 && cus < 0xffff && fit > 0) slots[dev].store(1u + (uint64_t)cus + ((uint64_t)fit << 16), std::memory_order_relaxed);
    }
    if (fit > 0 && per_cu > fit) per_cu = fit;              // a persistent grid must be resident at once to be worth anything
    const uint64_t resident = ((uint64_t)cus * (unsigned)per_cu + 7) / 8 * 8;          // a multiple of 8: a workgroup stays on its XCD
    if (resident >= 8 && resident < grid) grid = (unsigned)resident;
    return hipSuccess;
}

// Resident wavefronts per CU: 8 (two per SIMD) measured best or equal against 12 (what the 130 VGPRs allow) and 16
// (profiles/exp_columns_r05.txt, step 3); `waves` (the A/B tools' occupancy knob: waves per SIMD) overrides it.
constexpr int COLUMNS_WAVES_PER_CU = 8;
template <typename T, int SNK>
hipError_t launch_chain_columns_pipelined(const ChainArgs& a0, int waves, hipStream_t s)
{
    if (a0.n_blocks == 0) return hipSuccess;
    const int per_cu_override = waves > 0 ? 4 * waves : COLUMNS_WAVES_PER_CU;
    ChainArgs a = a0;
    constexpr unsigned TILE_BLOCKS = COLUMN_LANES_BPW;
    const uint64_t n_tiles = (a.n_blocks + TILE_BLOCKS - 1) / TILE_BLOCKS;
    a.tiles_per_xcd = (n_tiles + 7) / 8;
    if (a.tiles_per_xcd * 8 > 0x7fffffffull) return hipErrorInvalidValue;
    a.window_shift = tile_window_shift(SNK == SNK_ORIGINAL ? WIN_UNDELTA_PACK_UNTRANSPOSE : WIN_UNDELTA_PACK, WaveBlock<T>::TB, TILE_BLOCKS);
    if (a.widths) a.window_shift |= TILE_MAP_ROTATE;
    const unsigned lds = TILE_BLOCKS * (SNK == SNK_ORIGINAL && sizeof(T) == 1 ? ORIGINAL_U8_STRIDE : WaveBlock<T>::BLOCK_BYTES);
    unsigned grid = (unsigned)(a.tiles_per_xcd * 8);
    if (hipError_t e = persistent_grid<k_chain_columns_pipelined<T, SNK>>(lds, per_cu_override, grid); e != hipSuccess) return e;
    FL_LAUNCH((k_chain_columns_pipelined<T, SNK>), dim3(grid), dim3(64), lds, s, a);
    return hipGetLastError();
}

// ---- the ENCODE side on column lanes: transpose_delta_pack over a mixed-width u8 column, pipelined (round 5) ----------------------
//     pack::<widths[b]>(delta(transpose(block b), bases[b]))        (delta.rs:88-95 composed, per block)
// The lockstep kernel gathers the transposition byte by byte through LDS (1024 ds_read_u8 per block), takes the previous row from the
// lane group below by ds_bpermute and packs a block at a time with LDS atomics: LDS busy 0.55, 0.66-0.70 of the HBM peak
// (profiles/r05_sq_mixed_lockstep.txt).  Here, as in the decode kernel above, lane (j, c) owns cell column c of block first+j for all
// 8 rows: sixteen 8-byte reads + a register transpose give it its rows (runs_to_rows_u8), the row before is its own register, and
// the fields go into a per-lane bit buffer -- SPLIT like the decode's running sum: even and odd bytes in the 16-bit halves of two
// registers per word, where `(delta & mask) << fill` cannot run into a neighbouring element and a subtraction is one v_pk_sub_u16
// -- that emits one packed cell (word k of 16 FL lanes, macros.rs:84-92) whenever 8 bits are full.  The wavefront is persistent and
// software-pipelined exactly like the decode kernel: tile k+2's widths / offsets and tile k+1's values + bases in flight while tile k
// is encoded.
template <typename T> struct EncodeTile {
    u32x4 p[COLUMN_LANES_BPW][WaveBlock<T>::GROUPS];        // block j, original-order KiB g: this lane's 16 bytes
    Cell<T> base;
    unsigned wm;                                            // this lane's block's width (0 if the block is not encoded)
    unsigned wv;                                            // lane j: block first+j's width, 0 if it is not encoded
    uint64_t ov;                                            // lane j: its byte offset in the packed column
    uint64_t first;
};

template <typename T>
__device__ __forceinline__ void encode_issue(const ChainArgs& a, uint64_t first, unsigned count, const ColumnMeta& meta, unsigned lane,
                                             EncodeTile<T>& d, uint32_t& lane_errors)
{
    using G = WaveBlock<T>;
    constexpr unsigned BPW = COLUMN_LANES_BPW;
    const bool mixed = a.widths != nullptr;                 // wave-uniform
    const unsigned jm = lane >> 3;
    unsigned wv = a.width;
    uint64_t ov = (first + lane) * (uint64_t)(128u * a.width);
    uint32_t e = 0;
    if (mixed) {                                            // lane j checks block first+j (bitpacking.rs:93, :78-80), vector code
        wv = meta.wv;
        ov = meta.ov;
        const uint32_t misaligned = (ov & 15u) ? DEVERR_ALIGN : 0u;
        const uint32_t outside = (ov > a.packed_bytes || 128ull * wv > a.packed_bytes - ov) ? DEVERR_BOUNDS : 0u;
        e = wv > (unsigned)G::TB ? DEVERR_WIDTH : (misaligned | outside);
    }
    const bool mine = lane < count;
    e = mine ? e : 0u;
    lane_errors |= e;
    const bool okl = mine && e == 0;
    d.wv = okl ? wv : 0u;                                   // W == 0 writes nothing (macros.rs:52-53), like a block that is skipped
    d.ov = okl ? ov : 0ull;
    d.first = first;
    // the tile's 8 unpacked blocks are consecutive: one descriptor, 8 KiB, whatever the blocks' widths
    const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.in) + first * G::BLOCK_BYTES, 0, count * G::BLOCK_BYTES, 0x00020000);
    static_for<BPW>([&](auto Jt) {
        constexpr unsigned j = decltype(Jt)::value;
        static_for<G::GROUPS>([&](auto Gi) {
            constexpr unsigned g = decltype(Gi)::value;
            d.p[j][g] = __builtin_amdgcn_raw_buffer_load_b128(in_rs, lane * 16u + j * G::BLOCK_BYTES + g * 1024u, 0, 2 /* nt */);
        });
    });
    d.wm = (unsigned)__builtin_amdgcn_ds_bpermute((int)(jm * 4u), (int)d.wv);
    const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.bases) + first * 128u, 0, count * 128u, 0x00020000);
    d.base = __builtin_bit_cast(Cell<T>, __builtin_amdgcn_raw_buffer_load_b128(b_rs, lane * 16u, 0, 0));
}

__device__ __forceinline__ uint32_t pk_sub_u16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b)));
}

__device__ __forceinline__ void encode_consume_u8(const ChainArgs& a, const EncodeTile<uint8_t>& d, char* lds, unsigned lane)
{
    using T = uint8_t;
    using G = WaveBlock<T>;
    constexpr unsigned BPW = COLUMN_LANES_BPW, IMG = ORIGINAL_U8_STRIDE;
    const unsigned jm = lane >> 3, c16 = (lane & 7u) * 16u;
    static_for<BPW>([&](auto Jt) { *reinterpret_cast<u32x4*>(lds + decltype(Jt)::value * IMG + lane * 16u) = d.p[decltype(Jt)::value][0]; });
    wave_lds_fence();
    // transpose.rs:12-14: this lane's 16 FL lanes' runs of 8 rows -> the 8 rows' cells
    u32x2_t runs[16];
    const char* at = lds + jm * IMG + first_run_of_column(lane & 7u);
    static_for<16>([&](auto E) { runs[decltype(E)::value] = *reinterpret_cast<const u32x2_t*>(at + 64 * decltype(E)::value); });
    Cell<T> x[8];
    runs_to_rows_u8(runs, x);
    wave_lds_fence();                                       // every lane holds its rows: the images may be overwritten with packed cells
    // delta.rs:28-30 and macros.rs:72-92 per FL lane, in split form: ev / od = the even / odd bytes of a cell in the low bytes of u16x2
    const unsigned wm = d.wm;
    const uint32_t m = G::field_mask(wm);                   // (2^w - 1) in both halves; 0 for W == 0
    uint32_t pe[4], po[4], ae[4], ao[4];
    for (int k = 0; k < 4; ++k) { pe[k] = d.base.x[k]; po[k] = d.base.x[k] >> 8; ae[k] = 0; ao[k] = 0; }
    unsigned fill = 0;
    char* out_at = lds + jm * IMG + c16;                    // packed cell (word k, column c) of this lane's block: + 128 * k
    static_for<8>([&](auto R) {
        constexpr unsigned r = decltype(R)::value;
        for (int k = 0; k < 4; ++k) {
            const uint32_t ce = x[r].x[k], co = x[r].x[k] >> 8;
            const uint32_t de = pk_sub_u16(ce, pe[k]) & m, dd = pk_sub_u16(co, po[k]) & m;     // in[idx] - prev, masked to W bits (macros.rs:73)
            pe[k] = ce;
            po[k] = co;
            ae[k] |= de << fill;                            // fill + W <= 15: stays inside the 16-bit half
            ao[k] |= dd << fill;
        }
        fill += wm;
        if (fill >= 8u) {                                   // a word of this lane's 16 streams is complete (macros.rs:84-92)
            u32x4 cell;
            for (int k = 0; k < 4; ++k) {
                cell[k] = __builtin_amdgcn_perm(ao[k], ae[k], 0x06020400u);                    // od.b2 : ev.b2 : od.b0 : ev.b0
                ae[k] = (ae[k] >> 8) & 0x00ff00ffu;
                ao[k] = (ao[k] >> 8) & 0x00ff00ffu;
            }
            *reinterpret_cast<u32x4*>(out_at) = cell;
            out_at += 128;
            fill -= 8u;
        }
    });
    wave_lds_fence();
    // block j's 128 * w bytes leave 1 KiB-contiguously through a descriptor that ends where the block ends (nothing for W == 0 / a skipped block)
    static_for<BPW>([&](auto Jt) {
        constexpr unsigned j = decltype(Jt)::value;
        const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)d.wv, (int)j);
        const uint64_t packed_at = readlane_elem<uint64_t>(d.ov, j);
        const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(a.out + packed_at, 0, 128u * w, 0x00020000);
        const u32x4 v = *reinterpret_cast<const u32x4*>(lds + j * IMG + lane * 16u);
        __builtin_amdgcn_raw_buffer_store_b128(v, out_rs, lane * 16u, 0, STORE_AUX);
    });
    wave_lds_fence();
}

template <typename T>
__global__ __launch_bounds__(64) void k_chain_columns_encode_pipelined(ChainArgs a)
{
    static_assert(sizeof(T) == 1, "the register transpose and the split bit buffer are u8's");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr unsigned BPW = COLUMN_LANES_BPW;
    const unsigned tiles_per_xcd = (unsigned)a.tiles_per_xcd;                 // the launcher keeps 8 * tiles_per_xcd below 2^31
    const unsigned slots = tiles_per_xcd * 8, grid = gridDim.x;
    const unsigned n_tiles = (unsigned)((a.n_blocks + BPW - 1) / BPW);
    const unsigned lane = threadIdx.x;
    auto blocks_of = [&](unsigned t, uint64_t& first, unsigned& count) {      // slot t of the tile map; count = 0: nothing there
        const unsigned tile = t < slots ? xcd_tile<uint32_t>(t, tiles_per_xcd, a.window_shift) : n_tiles;
        const bool there = tile < n_tiles;
        first = there ? (uint64_t)tile * BPW : 0;
        const uint64_t left = a.n_blocks - first;
        count = there ? (left < BPW ? (unsigned)left : BPW) : 0u;
    };
    uint32_t lane_errors = 0;
    unsigned t1 = blockIdx.x + grid;
    uint64_t first0, first1;
    unsigned count0, count1;
    blocks_of(blockIdx.x, first0, count0);
    blocks_of(t1, first1, count1);
    const ColumnMeta m0 = columns_meta_load(a, first0, count0, lane);
    ColumnMeta m1 = columns_meta_load(a, first1, count1, lane);
    EncodeTile<T> cur;
    encode_issue<T>(a, first0, count0, m0, lane, cur, lane_errors);
    for (;;) {
        const unsigned t2 = t1 + grid;
        uint64_t first2;
        unsigned count2;
        blocks_of(t2, first2, count2);
        const ColumnMeta m2 = columns_meta_load(a, first2, count2, lane);      // tile k+2's metadata
        EncodeTile<T> nxt;
        encode_issue<T>(a, first1, count1, m1, lane, nxt, lane_errors);        // tile k+1's values and bases
        encode_consume_u8(a, cur, lds, lane);                                  // tile k
        if (t1 >= slots) break;
        cur = nxt;
        m1 = m2;
        first1 = first2;
        count1 = count2;
        t1 = t2;
    }
    if (__builtin_amdgcn_ballot_w64(lane_errors != 0)) {
        for (int o = 32; o; o >>= 1) lane_errors |= (uint32_t)__shfl_xor((int)lane_errors, o);
        raise_device_error(a.err_flag, lane_errors, lane);
    }
}

template <typename T>
hipError_t launch_chain_columns_encode_pipelined(const ChainArgs& a0, int waves, hipStream_t s)
{
    if (a0.n_blocks == 0) return hipSuccess;
    ChainArgs a = a0;
    constexpr unsigned TILE_BLOCKS = COLUMN_LANES_BPW;
    const uint64_t n_tiles = (a.n_blocks + TILE_BLOCKS - 1) / TILE_BLOCKS;
    a.tiles_per_xcd = (n_tiles + 7) / 8;
    if (a.tiles_per_xcd * 8 > 0x7fffffffull) return hipErrorInvalidValue;
    a.window_shift = tile_window_shift(WIN_TRANSPOSE_DELTA_PACK, WaveBlock<T>::TB, TILE_BLOCKS);
    if (a.widths) a.window_shift |= TILE_MAP_ROTATE;
    const unsigned lds = TILE_BLOCKS * ORIGINAL_U8_STRIDE;
    unsigned grid = (unsigned)(a.tiles_per_xcd * 8);
    if (hipError_t e = persistent_grid<k_chain_columns_encode_pipelined<T>>(lds, waves > 0 ? 4 * waves : COLUMNS_WAVES_PER_CU, grid); e != hipSuccess) return e;
    FL_LAUNCH((k_chain_columns_encode_pipelined<T>), dim3(grid), dim3(64), lds, s, a);
    return hipGetLastError();
}

// what the C ABI asks for
enum ChainOp { OP_UNDELTA_PACK = 0, OP_UNDELTA = 1, OP_DELTA = 2, OP_UNTRANSPOSE = 3, OP_TRANSPOSE = 4,
               OP_UNDELTA_PACK_UNTRANSPOSE = 5, OP_TRANSPOSE_DELTA_PACK = 6 };
template <typename T> chain_launch_t chain_launcher(int op);
// the mixed-width form (ChainArgs.widths != nullptr) of the three ops with a packed side
template <typename T> chain_launch_t chain_widths_launcher(int op);
// the two-blocks-per-wavefront form of an op (u32 / u64 undelta_pack), or nullptr
template <typename T> chain_launch_t chain_launcher_two_blocks(int op);

}  // namespace fl
