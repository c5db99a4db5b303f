// fl_consume.hpp -- fused consumers / producers' helpers around the codec.
// EXTENSIONS: the reference has no such functions; each is defined as a plain reduction over
// what the reference functions produce, so the oracle composition is the specification.
//   unpack_block_sums<W> : s[b] = sum_{i<1024} unpack::<W>(packed_b)[i]   (wrapping u64)
//                          -- the read-bound regime: 128*W bytes in, 8 bytes out per block.
//   block_min_max        : mn[b], mx[b] over the 1024 values of unpacked block b -- what an
//                          encoder needs to pick FoR's reference (min) and the width
//                          (bits(max - min)) before calling for_pack::<W> (ffor.rs:24-36).
#pragma once
#include "fl_kernels.hpp"
#include "fl_widths.hpp"

namespace fl {

// Waves per SIMD of the consumer kernels (same-buffer A/B of builds capped at 2..6 waves, profiles/abconsume_r03.txt).
// A thread keeps all W packed cells of its column in registers; uncapped, hipcc allocates ~62 VGPRs to reach 8 waves per
// SIMD and serialises the loads of the wider widths.  Narrow widths are VALU-bound (profiles/r03_pmc_sq_counters.csv: up to
// 87 % of the VALU issue slots at u16 W=3) and need every wave; from W = 6 up fewer, fatter waves with every load in flight
// stream better: compare +3...6 % (u64 W=56: +17 %), sums +2...6 % at W = 6..12.
constexpr int compare_max_waves(int w) { return w <= 5 ? 8 : w <= 9 ? 3 : 2; }
constexpr int sums_max_waves(int w) { return (w >= 6 && w <= 12) ? 3 : 8; }

struct ReduceArgs {
    const u32x4* in;
    void* out0;            // sums (uint64 per block) or mins (T per block)
    void* out1;            // maxs (T per block) or unused
    uint64_t n_blocks;
    uint64_t tiles_per_xcd;
};

__device__ __forceinline__ bool tile_of_workgroup(const ReduceArgs& a, uint64_t& tile)
{
    const uint64_t n_tiles = (a.n_blocks + BLOCKS_PER_WG - 1) / BLOCKS_PER_WG;
    tile = (uint64_t)(blockIdx.x & 7u) * a.tiles_per_xcd + (blockIdx.x >> 3);
    return tile < n_tiles;
}

// sum of the elements of one cell, as u64
template <typename T> __device__ __forceinline__ uint64_t cell_hsum(const Cell<T>& c)
{
    if constexpr (sizeof(T) == 8) return c.x[0] + c.x[1];
    else if constexpr (sizeof(T) == 4) return (uint64_t)c.x[0] + c.x[1] + c.x[2] + c.x[3];
    else if constexpr (sizeof(T) == 2) {
        uint32_t s = 0;
        for (int i = 0; i < 4; ++i) s = __builtin_amdgcn_sad_u16(c.x[i], 0u, s);     // v_sad_u16: both halfwords + s, one op
        return s;
    } else {
        uint32_t s = 0;
        for (int i = 0; i < 4; ++i) s = __builtin_amdgcn_sad_u8(c.x[i], 0u, s);      // v_sad_u8: all four bytes + s, one op
        return s;
    }
}

// reduce over the 8 threads (cell columns) of a block
__device__ __forceinline__ uint64_t group8_sum(uint64_t v)
{
    for (int m = 1; m < 8; m <<= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)v, m, 8), hi = __shfl_xor((uint32_t)(v >> 32), m, 8);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}
template <typename T> __device__ __forceinline__ T group8_minmax(T v, bool want_max)
{
    for (int m = 1; m < 8; m <<= 1) {
        T o;
        if constexpr (sizeof(T) == 8) {
            const uint32_t lo = __shfl_xor((uint32_t)v, m, 8), hi = __shfl_xor((uint32_t)(v >> 32), m, 8);
            o = ((uint64_t)hi << 32) | lo;
        } else {
            o = (T)__shfl_xor((uint32_t)v, m, 8);
        }
        v = want_max ? (o > v ? o : v) : (o < v ? o : v);
    }
    return v;
}

// Narrow widths are VALU-bound when every field is extracted (profiles/r03_pmc_sq_derived.txt: 0.80 of the issue rate at u16
// W=3).  The SUM of a block does not need the fields: bit p of an FL lane's stream is bit (p mod W) of field p / W
// (macros.rs:72-92), so
//     sum over all 1024 values = sum_{b < W} 2^b * popcount(packed bits whose stream position is = b mod W),
// and a packed 32-bit register holds stream positions  row*T + (bit mod T)  of 32/T lanes at once: one v_and + one
// accumulating v_bcnt per (register, class), 2*W operations per register = W*W/16 per value (0.56 at W=3 instead of 2.2).
// Pays for W <= 5; exact (the counts are small integers, the weighted sum fits 64 bits).
template <typename T, int W> constexpr bool sums_by_bit_class() { return W >= 1 && W <= 5 && W < (int)(sizeof(T) * 8); }

// 32-bit mask of the bits of packed row `row` (register half `half` of a u64 word) whose stream position is = b mod W
template <typename T, int W> constexpr uint32_t bit_class_mask(int row, int half, int b)
{
    constexpr int TB = (int)sizeof(T) * 8;
    uint32_t m = 0;
    for (int i = 0; i < 32; ++i) {
        const int j = TB >= 32 ? i + 32 * half : i % TB;          // bit inside the T-bit word
        if ((row * TB + j) % W == b) m |= 1u << i;
    }
    return m;
}

template <typename T, int W>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(1, sums_max_waves(W)))) void k_unpack_block_sums(ReduceArgs a)
{
    uint64_t tile;
    if (!tile_of_workgroup(a, tile)) return;
    const unsigned tid = threadIdx.x;
    const uint64_t blk = tile * BLOCKS_PER_WG + (tid >> 3);
    const unsigned c = tid & 7u;
    if (blk >= a.n_blocks) return;     // whole 8-thread groups leave together
    Cell<T> in[W ? W : 1];
    const u32x4* pk = a.in + blk * (uint64_t)(8 * W) + c;
    static_for<W>([&](auto Wd) { in[decltype(Wd)::value] = load_cell<T, true>(pk + 8 * decltype(Wd)::value); });
    uint64_t acc = 0;
    if constexpr (sums_by_bit_class<T, W>()) {
        uint32_t cnt[W];
        static_for<W>([&](auto B) { cnt[decltype(B)::value] = 0; });
        static_for<W>([&](auto Row) {
            constexpr int row = decltype(Row)::value;
            const u32x4 r = __builtin_bit_cast(u32x4, in[row]);
            static_for<4>([&](auto K) {
                constexpr int k = decltype(K)::value;
                static_for<W>([&](auto B) {
                    constexpr int b = decltype(B)::value;
                    constexpr uint32_t m = bit_class_mask<T, W>(row, sizeof(T) == 8 ? (k & 1) : 0, b);
                    cnt[b] += __builtin_popcount(r[k] & m);
                });
            });
        });
        static_for<W>([&](auto B) { acc += (uint64_t)cnt[decltype(B)::value] << decltype(B)::value; });
    } else {
        unpack_rows<T, W>(in, [&](auto, const Cell<T>& v) { acc += cell_hsum<T>(v); });
    }
    acc = group8_sum(acc);
    if (c == 0) static_cast<uint64_t*>(a.out0)[blk] = acc;
}

template <typename T>
__global__ __launch_bounds__(WG) void k_block_min_max(ReduceArgs a)
{
    constexpr int TB = Elem<T>::BITS;
    constexpr int N = Elem<T>::PER_CELL;
    uint64_t tile;
    if (!tile_of_workgroup(a, tile)) return;
    const unsigned tid = threadIdx.x;
    const uint64_t blk = tile * BLOCKS_PER_WG + (tid >> 3);
    const unsigned c = tid & 7u;
    if (blk >= a.n_blocks) return;
    const u32x4* un = a.in + blk * (uint64_t)Elem<T>::CELLS_PER_BLOCK + c;
    T mn = (T) ~(T)0, mx = 0;
    static_for<TB>([&](auto R) {
        // min/max do not care about the order of values, so rows are read in storage order
        const Cell<T> v = load_cell<T, true>(un + 8 * decltype(R)::value);
        static_for<N>([&](auto E) {
            const T x = (T)cell_get<T>(v, decltype(E)::value);
            mn = x < mn ? x : mn;
            mx = x > mx ? x : mx;
        });
    });
    mn = group8_minmax<T>(mn, false);
    mx = group8_minmax<T>(mx, true);
    if (c == 0) {
        static_cast<T*>(a.out0)[blk] = mn;
        static_cast<T*>(a.out1)[blk] = mx;
    }
}

// ---------------------------------------------------------------------------
// unpack_compare<W>: mask bit i of block b = cmp(unpack::<W>(packed_b)[i], constant), i in the
// unpacked (FastLanes index) order -- a selection vector straight from packed data: 128*W bytes
// in, 128 bytes out per block.  Every predicate is reduced on the host to one of two primitives
// (x == k, x <= k) plus a final complement.  Address-row j of a block is elements [j*LANES,
// (j+1)*LANES) = mask bits [j*LANES, (j+1)*LANES); every column thread contributes PER_CELL contiguous bits
// of each row.  The pieces are put together in a wave-private LDS image of the block's 128-byte mask
// (u8/u16/u32) or with DPP OR-reductions (u64), and the mask leaves as one coalesced 16-byte store per thread.
// ---------------------------------------------------------------------------
struct CompareArgs {
    const u32x4* in;
    u32x4* mask;           // [n_blocks][8] cells = 128 bytes per block
    uint64_t constant;     // k of the primitive, already adjusted
    uint32_t is_eq;        // 1: x == k, 0: x <= k (selects the kernel instance on the host)
    uint32_t invert;       // complement the result
    uint64_t n_blocks;
    uint64_t tiles_per_xcd;
};

__device__ __forceinline__ uint32_t or_allreduce8(uint32_t x)
{
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]: lane ^ 1
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]: lane ^ 2
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, true);   // row_half_mirror: lane -> 7 - lane
    return x;
}

// Per-row predicate bits of one cell column: bit e of the result = cmp(element e of the cell, k), e < PER_CELL.
//   u64 / u32 : one compare per element.
//   u16 / u8  : SWAR -- all elements of a 32-bit word are compared at once, the verdicts land in the top bit of
//               each element field (H), and the H bits of the cell's four words are squeezed together
//               (per-element extraction cost ~4.5 VALU operations per value and capped these kernels at 0.47 of the
//               HBM peak; this is ~2.5).
template <typename T, int W, bool IS_EQ>
__device__ __forceinline__ uint32_t row_predicate_bits(const Cell<T>& v, T k)
{
    constexpr int N = Elem<T>::PER_CELL;
    if constexpr (sizeof(T) >= 4) {
        uint32_t bits = 0;
        static_for<N>([&](auto E) {
            const T x = (T)cell_get<T>(v, decltype(E)::value);
            const uint32_t p = IS_EQ ? (x == k) : (x <= k);
            bits |= p << decltype(E)::value;
        });
        return bits;
    } else {
        constexpr uint32_t H = sizeof(T) == 2 ? 0x80008000u : 0x80808080u;
        constexpr uint32_t L = ~H;
        const uint32_t kr = Cell<T>::splat(k).x[0];
        uint32_t p[4];
        for (int i = 0; i < 4; ++i) {
            const uint32_t x = v.x[i];
            if constexpr (IS_EQ) {
                const uint32_t y = x ^ kr;
                p[i] = ~(((y & L) + L) | y) & H;                           // field == 0
            } else if constexpr (W < (int)(sizeof(T) * 8)) {
                // elements are < 2^W <= 2^(T-1): their top bit is clear, so x <= k  <=>  k's top bit | low(k) >= x
                p[i] = (((kr | H) - x) | kr) & H;
            } else {
                const uint32_t ge_low = (kr | H) - (x & L);                // H bit: low(k) >= low(x), no borrow across fields
                p[i] = ((~x & kr) | (~(x ^ kr) & ge_low)) & H;
            }
        }
        if constexpr (sizeof(T) == 2) {
            // H bits 15 / 31 of word i -> bits 2i / 16 + 2i, then fold the upper halfword in between
            const uint32_t q = (p[0] >> 15) | (p[1] >> 13) | (p[2] >> 11) | (p[3] >> 9);
            return (q & 0x55u) | ((q >> 15) & 0xAAu);
        } else {
            // H bits 7/15/23/31 of a word -> one nibble: (x * 0x01020408) >> 24 moves bit 8b to bit 24 + b (no two
            // partial products share a bit position, so there are no carries)
            uint32_t bits = 0;
            for (int i = 0; i < 4; ++i) bits |= (((p[i] >> 7) * 0x01020408u) >> 24) << (4 * i);
            return bits;
        }
    }
}

// The same verdict bits for one logical row straight from the PACKED words where that is cheaper: for u8 / u16, a field that
// lies inside one packed word below its top bit ((row*W) % T + W <= T - 1) is compared IN PLACE -- masked where it sits
// (one v_and instead of shift + and) against the constant shifted to the same position, the element's top bit serving as
// the SWAR guard (2 operations instead of 3).  k is clamped to the field's range first (x <= k is true for every W-bit x once
// k >= 2^W - 1; x == k is false for k >= 2^W), so the shifted constant always fits.  ~20 instead of ~32 VALU operations per
// row of a u16 column at the narrow widths, which are VALU-bound (profiles/r03_pmc_sq_derived.txt).
template <typename T, int W, int ROW, bool IS_EQ>
__device__ __forceinline__ uint32_t row_predicate_bits_of_row(const Cell<T>* in, T k)
{
    constexpr int TB = Elem<T>::BITS;
    constexpr int sh = W ? (ROW * W) % TB : 0;
    if constexpr (sizeof(T) <= 2 && W >= 1 && W < TB && sh + W <= TB - 1) {
        constexpr int word = (ROW * W) / TB;
        constexpr uint32_t H = sizeof(T) == 2 ? 0x80008000u : 0x80808080u;
        constexpr uint32_t L = ~H;
        constexpr T FM = (T)((1u << W) - 1u);
        constexpr uint32_t M = Cell<T>::rep(W) << sh;                     // the field where it sits, in every element
        const bool beyond = k > FM;                                       // wave-uniform
        const uint32_t ks = Cell<T>::splat((T)((beyond ? FM : k) << sh)).x[0];
        uint32_t p[4];
        for (int i = 0; i < 4; ++i) {
            const uint32_t x = in[word].x[i] & M;
            if constexpr (IS_EQ) {
                const uint32_t y = x ^ ks;                                // bits of the field only (top bit clear)
                p[i] = beyond ? 0u : (~((y + L) | y) & H);                // field == 0
            } else {
                p[i] = ((ks | H) - x) & H;                                // no borrow across elements: x, ks < 2^(T-1)
            }
        }
        if constexpr (sizeof(T) == 2) {
            const uint32_t q = (p[0] >> 15) | (p[1] >> 13) | (p[2] >> 11) | (p[3] >> 9);
            return (q & 0x55u) | ((q >> 15) & 0xAAu);
        } else {
            uint32_t bits = 0;
            for (int i = 0; i < 4; ++i) bits |= (((p[i] >> 7) * 0x01020408u) >> 24) << (4 * i);
            return bits;
        }
    } else {
        return row_predicate_bits<T, W, IS_EQ>(unpack_row<T, W, ROW>(in), k);
    }
}

// u64: the 8 column threads OR their 2-bit pieces together with three DPP steps per 32-bit mask word.
template <typename T, int W, bool IS_EQ>
__device__ __forceinline__ void compare_block_dpp(const Cell<T>* in, T k, unsigned c, uint32_t (&keep)[4])
{
    constexpr int TB = Elem<T>::BITS;
    constexpr int N = Elem<T>::PER_CELL;              // bits this thread contributes per row
    constexpr int LANES = Elem<T>::LANES;             // bits per address-row
    constexpr int PER_S = TB / 8;
    static_assert(LANES <= 32, "one or more address-rows per mask word");
    uint32_t piece[TB];
    static_for<TB>([&](auto J) {
        constexpr int j = decltype(J)::value;
        constexpr int row = fl_order((j % PER_S) * (8 / PER_S)) * 8 + j / PER_S;
        piece[j] = row_predicate_bits<T, W, IS_EQ>(unpack_row<T, W, row>(in), k);
    });
    static_for<32>([&](auto D) {
        constexpr int d = decltype(D)::value;
        constexpr int RPW = 32 / LANES;               // address-rows per mask word
        uint32_t w = 0;
        static_for<RPW>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            w |= piece[d * RPW + q] << (q * LANES + c * N);
        });
        w = or_allreduce8(w);
        if (d / 4 == (int)c) keep[d % 4] = w;
    });
}

// u8 / u16 / u32: the block's 128-byte mask is assembled in a wave-private LDS image -- address-row j is bits
// [j*LANES, (j+1)*LANES) and thread c owns PER_CELL contiguous bits of it (16 bits: ds_write_b16, 8 bits: ds_write_b8,
// 4 bits: one DPP exchange with the neighbour column makes a byte) -- and read back as one 16-byte cell per thread.
template <typename T, int W, bool IS_EQ>
__device__ __forceinline__ void compare_block_lds(const Cell<T>* in, T k, unsigned c, char* lds_blk, uint32_t (&keep)[4])
{
    constexpr int TB = Elem<T>::BITS;
    constexpr int PER_S = TB / 8;
    static_for<TB>([&](auto J) {
        constexpr int j = decltype(J)::value;
        constexpr int row = fl_order((j % PER_S) * (8 / PER_S)) * 8 + j / PER_S;
        const uint32_t bits = row_predicate_bits_of_row<T, W, row, IS_EQ>(in, k);
        if constexpr (sizeof(T) == 1) {
            *reinterpret_cast<uint16_t*>(lds_blk + j * 16 + c * 2) = (uint16_t)bits;
        } else if constexpr (sizeof(T) == 2) {
            *reinterpret_cast<uint8_t*>(lds_blk + j * 8 + c) = (uint8_t)bits;
        } else {
            uint32_t w = bits << (4 * (c & 1u));
            w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]: lane ^ 1
            *reinterpret_cast<uint8_t*>(lds_blk + j * 4 + (c >> 1)) = (uint8_t)w;           // both columns of the pair write the same byte
        }
    });
    wave_lds_fence();
    const u32x4 m = *reinterpret_cast<const u32x4*>(lds_blk + c * 16);
    keep[0] = m[0]; keep[1] = m[1]; keep[2] = m[2]; keep[3] = m[3];
}

template <typename T, int W, bool IS_EQ>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(1, compare_max_waves(W)))) void k_unpack_compare(CompareArgs a)
{
    const uint64_t n_tiles = (a.n_blocks + BLOCKS_PER_WG - 1) / BLOCKS_PER_WG;
    const uint64_t tile = (uint64_t)(blockIdx.x & 7u) * a.tiles_per_xcd + (blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    const unsigned tid = threadIdx.x;
    const uint64_t blk = tile * BLOCKS_PER_WG + (tid >> 3);
    const unsigned c = tid & 7u;
    if (blk >= a.n_blocks) return;     // whole 8-thread groups leave together (DPP / LDS exchange stays inside a group)
    Cell<T> in[W ? W : 1];
    const u32x4* pk = a.in + blk * (uint64_t)(8 * W) + c;
    static_for<W>([&](auto Wd) { in[decltype(Wd)::value] = load_cell<T, true>(pk + 8 * decltype(Wd)::value); });
    uint32_t keep[4] = {0, 0, 0, 0};
    if constexpr (sizeof(T) == 8) {
        compare_block_dpp<T, W, IS_EQ>(in, (T)a.constant, c, keep);
    } else {
        // 144-byte stride between the mask images of a wavefront's 8 blocks: at 128 bytes (= all 32 banks) the same byte
        // of every block falls into the same bank and each ds_write is an 8-way conflict (profiles/r03_pmc_sq_counters.csv:
        // SQ_LDS_BANK_CONFLICT = 0.71-0.74 of SQ_LDS_IDX_ACTIVE before); 36 dwords shift each block by 4 banks
        constexpr unsigned MASK_STRIDE = 144;
        __shared__ __attribute__((aligned(16))) char lds[BLOCKS_PER_WG * MASK_STRIDE];
        compare_block_lds<T, W, IS_EQ>(in, (T)a.constant, c, lds + (tid >> 3) * MASK_STRIDE, keep);
    }
    const uint32_t flip = a.invert ? ~0u : 0u;
    u32x4 out = {keep[0] ^ flip, keep[1] ^ flip, keep[2] ^ flip, keep[3] ^ flip};
    a.mask[blk * 8 + c] = out;
}

typedef hipError_t (*compare_launch_t)(const CompareArgs&, hipStream_t);
template <typename T, int W, bool IS_EQ> hipError_t launch_unpack_compare(const CompareArgs& a0, hipStream_t s)
{
    if (a0.n_blocks == 0) return hipSuccess;
    CompareArgs a = a0;
    const uint64_t n_tiles = (a.n_blocks + BLOCKS_PER_WG - 1) / BLOCKS_PER_WG;
    a.tiles_per_xcd = (n_tiles + 7) / 8;
    hipLaunchKernelGGL((k_unpack_compare<T, W, IS_EQ>), dim3((unsigned)(a.tiles_per_xcd * 8)), dim3(WG), 0, s, a);
    return hipGetLastError();
}
template <typename T> struct CompareTable { compare_launch_t fn[Elem<T>::BITS + 1]; };
template <typename T, bool IS_EQ, int... Ws>
constexpr CompareTable<T> make_compare_table(std::integer_sequence<int, Ws...>)
{
    return CompareTable<T>{{&launch_unpack_compare<T, Ws, IS_EQ>...}};
}
// the two primitives live in separate translation units (families 11 / 12) to build in parallel
template <typename T, bool IS_EQ> const CompareTable<T>& compare_table_impl();

// ---------------------------------------------------------------------------
// unpack_compare on the wave-per-block mapping (u32 / u64, runtime width): the packed block arrives in the wave's LDS image by
// non-temporal LDS-DMA (1 KiB-contiguous reads, fl_widths.hpp), lane (i, c) funnel-shifts the cell of address-row 8k+i,
// column c of every 1-KiB group k -- N = 16/sizeof(T) elements with CONSECUTIVE element indices k*1024/sizeof(T) + lane*N --
// so its N verdicts are N consecutive mask bits; the 8 lanes of a lane group (u64: 16 lanes) OR them into one 32-bit mask
// word with DPP steps, and 32 lanes store the block's 32 words.  More VALU per value than the cell-column kernel (the
// shift is a register, not a constant), but the wide widths are nowhere near VALU-bound there (0.26-0.38 of the issue
// rate) and read 1 KiB contiguous here instead of 8 x 128 B.
// ---------------------------------------------------------------------------
template <typename T, bool IS_EQ>
__global__ __launch_bounds__(WG) void k_compare_wave(CompareArgs a, unsigned w)
{
    static_assert(sizeof(T) >= 4, "SWAR types stay on the cell-column kernel");
    using G = WaveBlock<T>;
    constexpr int TB = G::TB;
    constexpr int N = 16 / (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char lds_all[];
    const uint64_t n_tiles = (a.n_blocks + (WG / 64) - 1) / (WG / 64);
    const uint64_t tile = (uint64_t)(blockIdx.x & 7u) * a.tiles_per_xcd + (blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    const unsigned tid = threadIdx.x;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
    const uint64_t blk = tile * (WG / 64) + wave;
    if (blk >= a.n_blocks) return;
    char* lds = lds_all + wave * G::BLOCK_BYTES;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.in)) + blk * (uint64_t)(128u * w), 0, 128u * w, 0x00020000);
    static_for<G::GROUPS>([&](auto Gi) {
        constexpr int g = decltype(Gi)::value;
        if (8u * g < w) dma_1k_to_lds<RD_DMA_NT, g * 1024>(rs, lds, lane);
    });
    wait_lds_dma();
    wave_lds_fence();
    const T k = (T)a.constant;
    const unsigned i = lane >> 3, c = lane & 7u, c16 = c * 16u;
    const typename G::word_t m = G::field_mask(w);
    unsigned bit = G::row_base(i) * w;
    const unsigned step = G::KSTEP * w;
    const unsigned last = (w - 1u) * 128u;
    uint32_t mine = 0;
    static_for<G::GROUPS>([&](auto K) {
        constexpr unsigned kk = decltype(K)::value;
        const unsigned word = bit >> G::LOG_TB, sh = bit & (TB - 1u);
        const unsigned a0 = word * 128u;
        const unsigned a1 = a0 + 128u < last ? a0 + 128u : last;            // macros.rs:156
        const Cell<T> cur = __builtin_bit_cast(Cell<T>, *reinterpret_cast<const u32x4*>(lds + a0 + c16));
        const Cell<T> nxt = __builtin_bit_cast(Cell<T>, *reinterpret_cast<const u32x4*>(lds + a1 + c16));
        const Cell<T> v = G::funnel(cur, nxt, sh, m);
        uint32_t bits = 0;
        static_for<N>([&](auto E) {
            const T x = (T)cell_get<T>(v, decltype(E)::value);
            const uint32_t p = IS_EQ ? (x == k) : (x <= k);
            bits |= p << decltype(E)::value;
        });
        // this cell's N elements are mask bits [kk*(1024/sizeof(T))... ] = element index kk*64*N + lane*N
        if constexpr (sizeof(T) == 4) {
            uint32_t wd = or_allreduce8(bits << (4u * c));                  // 8 lanes x 4 bits = mask word kk*8 + i
            if (c == kk) mine = wd;                                          // GROUPS = 4: lanes c < 4 keep a word each
        } else {
            uint32_t wd = or_allreduce8(bits << (2u * c + 16u * (i & 1u)));  // 8 lanes x 2 bits in the lane group's half ...
            wd |= (uint32_t)__shfl_xor((int)wd, 8, 64);                      // ... + the neighbouring group = mask word kk*4 + i/2
            if (c == kk) mine = wd;                                          // GROUPS = 8: lanes (i even, c) keep a word each
        }
        bit += step;
    });
    const uint32_t flip = a.invert ? ~0u : 0u;
    uint32_t* out = reinterpret_cast<uint32_t*>(a.mask) + blk * 32u;
    if constexpr (sizeof(T) == 4) {
        if (c < 4u) out[c * 8u + i] = mine ^ flip;
    } else {
        if ((i & 1u) == 0u) out[c * 4u + (i >> 1)] = mine ^ flip;
    }
}

typedef hipError_t (*compare_wave_launch_t)(const CompareArgs&, unsigned w, int waves, hipStream_t);
template <typename T, bool IS_EQ> hipError_t launch_compare_wave(const CompareArgs& a0, unsigned w, int waves, hipStream_t s)
{
    if (a0.n_blocks == 0) return hipSuccess;
    if constexpr (sizeof(T) >= 4) {
        CompareArgs a = a0;
        const uint64_t n_tiles = (a.n_blocks + (WG / 64) - 1) / (WG / 64);
        a.tiles_per_xcd = (n_tiles + 7) / 8;
        if (a.tiles_per_xcd * 8 > 0x7fffffffull) return hipErrorInvalidValue;
        hipLaunchKernelGGL((k_compare_wave<T, IS_EQ>), dim3((unsigned)(a.tiles_per_xcd * 8)), dim3(WG), widths_lds_bytes<T>(waves), s, a, w);
        return hipGetLastError();
    } else {
        return hipErrorInvalidValue;
    }
}
// nullptr for the SWAR types (u8 / u16)
template <typename T, bool IS_EQ> compare_wave_launch_t compare_wave_launcher();

typedef hipError_t (*reduce_launch_t)(const ReduceArgs&, hipStream_t);

inline unsigned plan_grid(ReduceArgs& a)
{
    const uint64_t n_tiles = (a.n_blocks + BLOCKS_PER_WG - 1) / BLOCKS_PER_WG;
    a.tiles_per_xcd = (n_tiles + 7) / 8;
    return (unsigned)(a.tiles_per_xcd * 8);
}
template <typename T, int W> hipError_t launch_unpack_block_sums(const ReduceArgs& a0, hipStream_t s)
{
    if (a0.n_blocks == 0) return hipSuccess;
    ReduceArgs a = a0;
    const unsigned grid = plan_grid(a);
    hipLaunchKernelGGL((k_unpack_block_sums<T, W>), dim3(grid), dim3(WG), 0, s, a);
    return hipGetLastError();
}
template <typename T> hipError_t launch_block_min_max(const ReduceArgs& a0, hipStream_t s)
{
    if (a0.n_blocks == 0) return hipSuccess;
    ReduceArgs a = a0;
    const unsigned grid = plan_grid(a);
    hipLaunchKernelGGL((k_block_min_max<T>), dim3(grid), dim3(WG), 0, s, a);
    return hipGetLastError();
}

template <typename T> struct ReduceTable { reduce_launch_t fn[Elem<T>::BITS + 1]; };
template <typename T, int... Ws>
constexpr ReduceTable<T> make_sum_table(std::integer_sequence<int, Ws...>)
{
    return ReduceTable<T>{{&launch_unpack_block_sums<T, Ws>...}};
}
template <typename T> const ReduceTable<T>& sum_table_impl();
template <typename T> reduce_launch_t min_max_launcher();

}  // namespace fl
