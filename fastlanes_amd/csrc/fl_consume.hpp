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

namespace fl {

// Waves per SIMD of the consumer kernels (same-buffer A/B of builds capped at 2..6 waves, profiles/abconsume_r03.txt).
// A thread keeps all W packed cells of its column in registers; uncapped, hipcc allocates ~62 VGPRs to reach 8 waves per
// SIMD and serialises the loads of the wider widths.  Narrow widths are VALU-bound (profiles/r03_pmc_sq_counters.csv: up to
// 87 % of the VALU issue slots at u16 W=3) and need every wave; from W = 6 up fewer, fatter waves with every load in flight
// stream better: compare +3...6 % (u64 W=56: +17 %), sums +2...6 % at W = 6..12.
constexpr int compare_max_waves(int w) { return w <= 5 ? 8 : w <= 9 ? 3 : 2; }
// waves per SIMD the compare kernels are LAUNCHED at (0 = the cap above), see launch_unpack_compare
// Round 4 (profiles/abcompare_thin_r04.txt, three boxes, same buffers): once the narrow widths stopped being VALU-bound, fewer
// resident waves stream better for the narrow TYPES as well -- u8 W=3 at 4 waves 0.803 -> 0.865 / 0.803 -> 0.842 / a tie, u16 W=3 at
// 6 waves 0.798 -> 0.830 / 0.758 -> 0.765 -- as a bare stream of the same bytes does (4 waves: +2...4 %).
inline int compare_launch_waves(unsigned type_bits, unsigned w) { return w > 5 ? 0 : type_bits == 8 ? 4 : type_bits == 16 ? 6 : 0; }
constexpr int sums_max_waves(int w) { return (w >= 6 && w <= 12) ? 3 : 8; }

struct ReduceArgs {
    const u32x4* in;
    void* out0;            // sums (uint64 per block) or mins (T per block)
    void* out1;            // maxs (T per block) or unused
    uint64_t n_blocks;
    uint64_t tiles_per_xcd;
    unsigned window_shift;  // tile-map window (fl_kernels.hpp: xcd_tile)
};

__device__ __forceinline__ bool tile_of_workgroup(const ReduceArgs& a, uint64_t& tile)
{
    const uint64_t n_tiles = (a.n_blocks + BLOCKS_PER_WG - 1) / BLOCKS_PER_WG;
    tile = xcd_tile(blockIdx.x, a.tiles_per_xcd, a.window_shift);
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
// of each row.  The pieces are put together in a wave-private LDS image of the block's 128-byte mask (u8 / u16) or by a
// register butterfly between the 8 threads (u32 / u64), and the mask leaves as one coalesced 16-byte store per thread.
// ---------------------------------------------------------------------------
struct CompareArgs {
    const u32x4* in;
    u32x4* mask;           // [n_blocks][8] cells = 128 bytes per block
    uint64_t constant;     // k of the primitive, already adjusted
    uint32_t is_eq;        // 1: x == k, 0: x <= k (selects the kernel instance on the host)
    uint32_t invert;       // complement the result
    uint64_t n_blocks;
    uint64_t tiles_per_xcd;
    unsigned window_shift;  // tile-map window (fl_kernels.hpp: xcd_tile)
};

// u8 / u16 compare SWAR-wise: all elements of a 32-bit word at once, the verdict of an element landing in ONE bit of its
// field (bit G of every T-bit element).  squeeze_bit<T, G> gathers those bits of the cell's four words into PER_CELL
// contiguous bits, element e's verdict at bit e.  Round 4: these kernels are VALU-bound at the narrow widths
// (profiles/r03_pmc_sq_derived.txt: 1.06 of the nominal issue rate at u16 W=3) and a third of their instructions squeezed
// verdict bits together with shifts, ORs and (u8) a quarter-rate 32-bit multiply.  v_dot4_u32_u8 does it instead: a byte that
// is (1 << g) or 0 times a weight 2^e, four bytes per instruction, accumulating -- sum = (the verdict bits) << g.
//   u16: one v_perm_b32 picks the byte holding bit G of each of four elements (two words), one v_and isolates the bit, one
//        dot4 per four elements, one shift per row: 7 operations per row of 8 elements (before: ~10, plus a v_and per word)
//   u8 : one v_and + one dot4 per word, two accumulators (the weights are bytes: 2^0 .. 2^7), shift + v_lshl_or: 10 per row of
//        16 elements (before: 16, four of them quarter-rate multiplies)
// Whatever else the words hold is ignored.
template <typename T, int G>
__device__ __forceinline__ uint32_t squeeze_bit(const uint32_t (&t)[4])
{
    static_assert(sizeof(T) <= 2 && G >= 0 && G < (int)(sizeof(T) * 8), "");
    constexpr int b = G & 7;
    constexpr uint32_t m = 0x01010101u << b;
    if constexpr (sizeof(T) == 2) {
        constexpr uint32_t sel = G >= 8 ? 0x07050301u : 0x06040200u;    // bytes 1 / 3 or 0 / 2 of the two words
        const uint32_t lo4 = __builtin_amdgcn_perm(t[1], t[0], sel) & m;   // elements 0, 1, 2, 3
        const uint32_t hi4 = __builtin_amdgcn_perm(t[3], t[2], sel) & m;   // elements 4, 5, 6, 7
        uint32_t acc = __builtin_amdgcn_udot4(lo4, 0x08040201u, 0u, false);
        acc = __builtin_amdgcn_udot4(hi4, 0x80402010u, acc, false);
        return b ? acc >> b : acc;
    } else {
        uint32_t lo = __builtin_amdgcn_udot4(t[0] & m, 0x08040201u, 0u, false);
        lo = __builtin_amdgcn_udot4(t[1] & m, 0x80402010u, lo, false);
        uint32_t hi = __builtin_amdgcn_udot4(t[2] & m, 0x08040201u, 0u, false);
        hi = __builtin_amdgcn_udot4(t[3] & m, 0x80402010u, hi, false);
        return (b ? lo >> b : lo) | (hi << (8 - b));
    }
}

// Per-row predicate bits of one cell column: bit e of the result = cmp(element e of the cell, k), e < PER_CELL.
//   u64 / u32 : one compare per element (only used by the W = 0 form; compare_block_butterfly is their kernel).
//   u16 / u8  : SWAR -- the verdicts land in the top bit of each element and are squeezed together.
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
                p[i] = ~(((y & L) + L) | y);                               // top bit: field == 0
            } else if constexpr (W < (int)(sizeof(T) * 8)) {
                // elements are < 2^W <= 2^(T-1): their top bit is clear, so x <= k  <=>  k's top bit | low(k) >= x
                p[i] = ((kr | H) - x) | kr;
            } else {
                const uint32_t ge_low = (kr | H) - (x & L);                // H bit: low(k) >= low(x), no borrow across fields
                p[i] = (~x & kr) | (~(x ^ kr) & ge_low);
            }
        }
        return squeeze_bit<T, (int)(sizeof(T) * 8) - 1>(p);
    }
}

// Which rows of a u8 / u16 column are compared IN PLACE, i.e. in the packed domain, and how (round 4).
// A field that lies inside one packed word below the word's top bit ((row*W) % T + W <= T - 1) needs no extraction: masked
// where it sits and subtracted from [guard bit | k at the same position], the guard bit -- the bit just above the field --
// survives iff field <= k (a borrow-free SWAR compare; for ==: guard - (field ^ k) keeps the guard iff the fields are equal).
// The fields of consecutive rows are adjacent in a word, so every OTHER in-place field of a word can share one subtraction: the
// fields of the rows in between are masked away and their lowest bits serve as the guards.  A word's in-place rows therefore
// fall into two classes (even / odd rank inside the word), and one class costs 2 operations per 32-bit register (== : 3) for ALL
// its rows -- at u16 W=3 six classes cover 13 of the 16 rows: 48 operations where round 3 spent 104 (one v_and + one v_sub per
// row and register) -- then each row's verdict bits are squeezed out of the class's registers (squeeze_bit at the row's guard).
template <typename T, int W> struct InPlaceRows {
    static constexpr int TB = Elem<T>::BITS;
    static constexpr bool in_place(int r) { return W >= 1 && W < TB && (r * W) % TB + W <= TB - 1; }
    static constexpr int word(int r) { return (r * W) / TB; }
    static constexpr int shift(int r) { return (r * W) % TB; }
    // class of an in-place row: parity of its rank among the in-place rows of its word
    static constexpr int cls(int r)
    {
        int n = 0;
        for (int q = 0; q < r; ++q) n += (in_place(q) && word(q) == word(r)) ? 1 : 0;
        return n & 1;
    }
    static constexpr bool member(int r, int wd, int c) { return in_place(r) && word(r) == wd && cls(r) == c; }
    static constexpr bool any(int wd, int c)
    {
        for (int r = 0; r < TB; ++r) if (member(r, wd, c)) return true;
        return false;
    }
    // per element: the class's field bits / the lowest bit of every field / the guard bits; `rep` spreads over the elements of a word
    static constexpr uint32_t rep(uint32_t e) { return e * (sizeof(T) == 2 ? 0x00010001u : 0x01010101u); }
    static constexpr uint32_t fields(int wd, int c)
    {
        uint32_t m = 0;
        for (int r = 0; r < TB; ++r) if (member(r, wd, c)) m |= ((1u << W) - 1u) << shift(r);
        return rep(m);
    }
    static constexpr uint32_t ones(int wd, int c)
    {
        uint32_t m = 0;
        for (int r = 0; r < TB; ++r) if (member(r, wd, c)) m |= 1u << shift(r);
        return rep(m);
    }
    static constexpr uint32_t guards(int wd, int c) { return ones(wd, c) << W; }
    // address-row of logical row r (the inverse of WaveRowStore<T>::row_at; FL_ORDER is its own inverse, lib.rs:53-59)
    static constexpr int address_row(int r) { return (r % 8) * (TB / 8) + fl_order(r / 8) * (TB / 8) / 8; }
};

// u32 / u64: the verdict bits never leave the registers.
//   * One verdict costs two VALU operations: v_cmp writes the lane's verdict to VCC and v_addc_co (bits + bits + carry-in)
//     shifts it into the low end of a 32-bit word (shift_in_verdicts).  hipcc has no pattern for that (it emits v_cmp, v_cndmask, v_or / v_lshl:
//     2.75 per value), hence the two-instruction asm.
//   * A column thread's 128 verdicts (T rows x N elements) fill 4 words; what it must store is the 128 mask bits of ITS
//     T/8 address-rows over all 8 columns.  That is an 8 x 8 transpose of N*T/8-bit pieces between the 8 threads of a block,
//     done as a 3-step butterfly on words laid out for it: a word's bit positions are [.. | J2 J1 J0 | e] where J = the
//     address-row's owner (j / (T/8)) and e the element inside the cell; step i swaps "owner bit i" with "column bit i" --
//     a thread keeps the groups whose J_i equals bit i of its own column, takes the partner's (column ^ (1 << i)) groups
//     with the same J_i and puts them where the groups it gave away were (a rotate by the group size, one v_bfi).  After the
//     three steps bit positions read [.. | c2 c1 c0 | e]: the mask's own layout.  3 operations per word and step.
//     (Before: three DPP OR-steps per mask word for u64, a DPP exchange + ds_write_b8 per row and an LDS round trip for
//     u32: 176 and 270 of the 660 / 850 VALU operations per wavefront at u32 W=7 / u64 W=17, now 36 + 9.)
//   * DPP reaches lane ^ 1, lane ^ 2 and lane ^ 7 (row_half_mirror) of an 8-lane group in one operation, not lane ^ 4.  So
//     lanes 4..7 of a group take the columns in reverse (column_of_lane): then lane ^ 7 IS column ^ 4.  The 8 lanes still read
//     and write the same 128 contiguous bytes per row.
__device__ __forceinline__ unsigned column_of_lane(unsigned lane8) { return lane8 ^ ((lane8 & 4u) ? 3u : 0u); }

#define FL_CMP_ADDC(op, x) "v_cmp_" op " vcc, %[k], %[" x "]\n\tv_addc_co_u32 %[b], vcc, %[b], %[b], vcc\n\t"
// verdicts of x3, x2, x1, x0 (u32: the four elements of a cell) in that order; one asm statement per cell because hipcc puts an
// s_nop between two adjacent asm statements that touch VCC
template <bool IS_EQ> __device__ __forceinline__ void shift_in_verdicts(uint32_t& bits, uint32_t x3, uint32_t x2, uint32_t x1, uint32_t x0, uint32_t k)
{
    if constexpr (IS_EQ)
        asm(FL_CMP_ADDC("eq_u32", "x3") FL_CMP_ADDC("eq_u32", "x2") FL_CMP_ADDC("eq_u32", "x1") FL_CMP_ADDC("eq_u32", "x0")
            : [b] "+&v"(bits) : [x3] "v"(x3), [x2] "v"(x2), [x1] "v"(x1), [x0] "v"(x0), [k] "s"(k) : "vcc");
    else
        asm(FL_CMP_ADDC("ge_u32", "x3") FL_CMP_ADDC("ge_u32", "x2") FL_CMP_ADDC("ge_u32", "x1") FL_CMP_ADDC("ge_u32", "x0")     // k >= x
            : [b] "+&v"(bits) : [x3] "v"(x3), [x2] "v"(x2), [x1] "v"(x1), [x0] "v"(x0), [k] "s"(k) : "vcc");
}
template <bool IS_EQ> __device__ __forceinline__ void shift_in_verdicts(uint32_t& bits, uint32_t x1, uint32_t x0, uint32_t k)
{
    if constexpr (IS_EQ) asm(FL_CMP_ADDC("eq_u32", "x1") FL_CMP_ADDC("eq_u32", "x0") : [b] "+&v"(bits) : [x1] "v"(x1), [x0] "v"(x0), [k] "s"(k) : "vcc");
    else asm(FL_CMP_ADDC("ge_u32", "x1") FL_CMP_ADDC("ge_u32", "x0") : [b] "+&v"(bits) : [x1] "v"(x1), [x0] "v"(x0), [k] "s"(k) : "vcc");
}
template <bool IS_EQ> __device__ __forceinline__ void shift_in_verdicts(uint32_t& bits, uint64_t x1, uint64_t x0, uint64_t k)
{
    if constexpr (IS_EQ) asm(FL_CMP_ADDC("eq_u64", "x1") FL_CMP_ADDC("eq_u64", "x0") : [b] "+&v"(bits) : [x1] "v"(x1), [x0] "v"(x0), [k] "s"(k) : "vcc");
    else asm(FL_CMP_ADDC("ge_u64", "x1") FL_CMP_ADDC("ge_u64", "x0") : [b] "+&v"(bits) : [x1] "v"(x1), [x0] "v"(x0), [k] "s"(k) : "vcc");
}
#undef FL_CMP_ADDC

// the partner's word: column ^ (1 << STEP) (see column_of_lane for STEP = 2)
template <int STEP> __device__ __forceinline__ uint32_t butterfly_partner(uint32_t x)
{
    constexpr int ctrl = STEP == 0 ? 0xB1 /* quad_perm [1,0,3,2] */ : STEP == 1 ? 0x4E /* quad_perm [2,3,0,1] */ : 0x141 /* row_half_mirror */;
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, 0xF, 0xF, true);
}

// (address-row j, element e) whose verdict sits at bit b of pre-butterfly word m:  u32: word = j % 4, bit = (j / 4) * 4 + e;
// u64: word = (j / 2) % 4, bit = (j % 2) * 16 + (j / 8) * 2 + e  -- the owner j / (T/8) in the three bits above e
template <typename T> constexpr int butterfly_row(int m, int b) { return sizeof(T) == 4 ? 4 * (b >> 2) + m : 8 * ((b >> 1) & 7) + 2 * m + (b >> 4); }
template <typename T> constexpr int butterfly_elem(int b) { return sizeof(T) == 4 ? (b & 3) : (b & 1); }

// x <= k without extracting x: the field of logical row ROW, element ELEM, moved to the TOP of a 32-bit register -- one shift if it
// lies inside a dword of the lane's stream, one v_alignbit_b32 if it crosses into the next -- with whatever lay below it left
// in place.  [field | junk] <= [k | all ones] <=> field <= k, so the mask (v_and / v_bfe) of macros.rs:150-164 is not needed.
// 1 <= W <= 32 (for u64: the two dwords of a word are consecutive dwords of the stream).
template <typename T, int I, int ELEM> __device__ __forceinline__ uint32_t stream_dword(const Cell<T>* in)
{
    if constexpr (sizeof(T) == 4) return in[I].x[ELEM];
    else return __builtin_bit_cast(u32x4, in[I / 2])[2 * ELEM + I % 2];   // (as registers: a 64-bit shift would be re-derived as v_alignbit + v_and)
}
template <typename T, int W, int ROW, int ELEM> __device__ __forceinline__ uint32_t top_aligned_field(const Cell<T>* in)
{
    static_assert(sizeof(T) >= 4 && W >= 1 && W <= 32, "");
    constexpr int d = (ROW * W) / 32, s = (ROW * W) % 32;
    if constexpr (s + W == 32) return stream_dword<T, d, ELEM>(in);
    else if constexpr (s + W < 32) return stream_dword<T, d, ELEM>(in) << (32 - s - W);
    else return __builtin_amdgcn_alignbit(stream_dword<T, d + 1, ELEM>(in), stream_dword<T, d, ELEM>(in), s + W - 32);
}

template <typename T, int W, bool IS_EQ>
__device__ __forceinline__ void compare_block_butterfly(const Cell<T>* in, T k, unsigned c, uint32_t (&keep)[4])
{
    static_assert(sizeof(T) >= 4, "u8 / u16 compare SWAR-wise (compare_block_lds)");
    constexpr int TB = Elem<T>::BITS;
    constexpr int N = Elem<T>::PER_CELL;
    constexpr int LOG_N = N == 4 ? 2 : 1;
    constexpr int PER_S = TB / 8;
    // every value is < 2^W: x <= k <=> x <= min(k, 2^W - 1), and x == k is false beyond 2^W - 1; for W <= 32 the compare is then
    // a 32-bit one for u64 too
    constexpr T FM = W >= TB ? (T) ~(T)0 : (T)(((T)1 << (W % TB)) - 1);
    const bool beyond = k > FM;                                   // wave-uniform
    const T kc = beyond ? FM : k;
    constexpr int TOP = W >= 1 && W < 32 ? 32 - W : 0;
    const uint32_t k_top = ((uint32_t)kc << TOP) | ((1u << TOP) - 1u);   // [k | all ones], see top_aligned_field
    static_for<4>([&](auto M) {
        constexpr int m = decltype(M)::value;
        uint32_t bits = 0;
        static_for<32 / N>([&](auto Q) {
            constexpr int b0 = 32 - N * (decltype(Q)::value + 1);   // descending: the first verdict shifted in ends up on top
            constexpr int j = butterfly_row<T>(m, b0);              // bits b0 .. b0 + N - 1 are the N elements of one address-row
            static_assert(butterfly_elem<T>(b0) == 0 && butterfly_row<T>(m, b0 + N - 1) == j, "");
            constexpr int row = fl_order((j % PER_S) * (8 / PER_S)) * 8 + j / PER_S;
            if constexpr (!IS_EQ && W >= 1 && W <= 32) {
                if constexpr (N == 4)
                    shift_in_verdicts<false>(bits, top_aligned_field<T, W, row, 3>(in), top_aligned_field<T, W, row, 2>(in),
                                             top_aligned_field<T, W, row, 1>(in), top_aligned_field<T, W, row, 0>(in), k_top);
                else
                    shift_in_verdicts<false>(bits, top_aligned_field<T, W, row, 1>(in), top_aligned_field<T, W, row, 0>(in), k_top);
            } else {
                const Cell<T> v = unpack_row<T, W, row>(in);
                if constexpr (N == 4) shift_in_verdicts<IS_EQ>(bits, v.x[3], v.x[2], v.x[1], v.x[0], (uint32_t)kc);
                else if constexpr (W <= 32) shift_in_verdicts<IS_EQ>(bits, (uint32_t)v.x[1], (uint32_t)v.x[0], (uint32_t)kc);
                else shift_in_verdicts<IS_EQ>(bits, (uint64_t)v.x[1], (uint64_t)v.x[0], (uint64_t)kc);
            }
        });
        keep[m] = bits;
    });
    static_for<3>([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr int g = N << i;                                 // group size in bits at this step
        constexpr uint32_t LOW = g == 2 ? 0x33333333u : g == 4 ? 0x0F0F0F0Fu : g == 8 ? 0x00FF00FFu : 0x0000FFFFu;   // position bit (LOG_N + i) clear
        static_assert(LOG_N + i <= 4, "groups stay inside a word");
        const bool up = (c >> i) & 1u;
        const uint32_t mine = up ? (LOW << g) : LOW;              // the groups this thread keeps
        const uint32_t rot = up ? (uint32_t)g : 32u - g;          // received groups move down (up: their J_i = 1 slot -> 0) or up
        static_for<4>([&](auto M) {
            constexpr int m = decltype(M)::value;
            const uint32_t t = butterfly_partner<i>(keep[m]);
            keep[m] = (keep[m] & mine) | (__builtin_amdgcn_alignbit(t, t, rot) & ~mine);
        });
    });
    if (IS_EQ && beyond) keep[0] = keep[1] = keep[2] = keep[3] = 0u;
}

// u8 / u16: the block's 128-byte mask is assembled in a wave-private LDS image -- address-row j is bits
// [j*LANES, (j+1)*LANES) and thread c owns PER_CELL contiguous bits of it (16 bits: ds_write_b16, 8 bits: ds_write_b8)
// -- and read back as one 16-byte cell per thread.  Rows whose field lies inside one packed word are compared in place, a
// class of them per subtraction (InPlaceRows); the others (straddling fields, fields touching a word's top bit, W = 0, W = T)
// are extracted first (unpack_row, macros.rs:144-164).
template <typename T, int W, bool IS_EQ>
__device__ __forceinline__ void compare_block_lds(const Cell<T>* in, T k, unsigned c, char* lds_blk, uint32_t (&keep)[4])
{
    using P = InPlaceRows<T, W>;
    constexpr int TB = Elem<T>::BITS;
    auto put = [&](auto J, uint32_t bits) {
        constexpr int j = decltype(J)::value;
        static_assert(j >= 0 && j < TB, "");
        if constexpr (sizeof(T) == 1) *reinterpret_cast<uint16_t*>(lds_blk + j * 16 + c * 2) = (uint16_t)bits;
        else *reinterpret_cast<uint8_t*>(lds_blk + j * 8 + c) = (uint8_t)bits;
    };
    // every value is < 2^W: x <= k <=> x <= min(k, 2^W - 1); x == k is false beyond 2^W - 1 (the kernel clears the mask then)
    constexpr T FM = W >= TB ? (T) ~(T)0 : (T)(((uint32_t)1 << (W % TB)) - 1u);
    const uint32_t kc = k > FM ? FM : k;                              // wave-uniform (scalar)
    static_for<(W ? W : 1)>([&](auto WD) {
        constexpr int wd = decltype(WD)::value;
        static_for<2>([&](auto C) {
            constexpr int cl = decltype(C)::value;
            if constexpr (P::any(wd, cl)) {
                // (constexpr variables: a constexpr function called in a runtime expression is compiled as a runtime loop)
                constexpr uint32_t M = P::fields(wd, cl), G = P::guards(wd, cl), ONES = P::ones(wd, cl);
                const uint32_t ks = kc * ONES;                            // k at every field of the class (scalar; no carries: k < 2^W)
                uint32_t t[4];
                for (int i = 0; i < 4; ++i) {
                    if constexpr (IS_EQ) t[i] = G - ((in[wd].x[i] ^ ks) & M);      // guard survives iff field ^ k == 0
                    else t[i] = (ks | G) - (in[wd].x[i] & M);                      // guard survives iff field <= k
                }
                static_for<TB>([&](auto R) {
                    constexpr int r = decltype(R)::value;
                    static_assert(WaveRowStore<T>::row_at(P::address_row(r)) == r, "address_row inverts row_at");
                    if constexpr (P::member(r, wd, cl))
                        put(std::integral_constant<int, P::address_row(r)>{}, squeeze_bit<T, P::shift(r) + W>(t));
                });
            }
        });
    });
    static_for<TB>([&](auto R) {
        constexpr int r = decltype(R)::value;
        if constexpr (!P::in_place(r))
            put(std::integral_constant<int, P::address_row(r)>{}, row_predicate_bits<T, W, IS_EQ>(unpack_row<T, W, r>(in), (T)kc));
    });
    wave_lds_fence();
    const u32x4 m = *reinterpret_cast<const u32x4*>(lds_blk + c * 16);
    keep[0] = m[0]; keep[1] = m[1]; keep[2] = m[2]; keep[3] = m[3];
    if (IS_EQ && k > FM) keep[0] = keep[1] = keep[2] = keep[3] = 0u;   // wave-uniform
}

template <typename T, int W, bool IS_EQ>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(1, compare_max_waves(W)))) void k_unpack_compare(CompareArgs a)
{
    const uint64_t n_tiles = (a.n_blocks + BLOCKS_PER_WG - 1) / BLOCKS_PER_WG;
    const uint64_t tile = xcd_tile(blockIdx.x, a.tiles_per_xcd, a.window_shift);
    if (tile >= n_tiles) return;
    const unsigned tid = threadIdx.x;
    const uint64_t blk = tile * BLOCKS_PER_WG + (tid >> 3);
    const unsigned c = sizeof(T) >= 4 ? column_of_lane(tid & 7u) : (tid & 7u);
    if (blk >= a.n_blocks) return;     // whole 8-thread groups leave together (DPP / LDS exchange stays inside a group)
    Cell<T> in[W ? W : 1];
    const u32x4* pk = a.in + blk * (uint64_t)(8 * W) + c;
    static_for<W>([&](auto Wd) { in[decltype(Wd)::value] = load_cell<T, true>(pk + 8 * decltype(Wd)::value); });
    uint32_t keep[4] = {0, 0, 0, 0};
    if constexpr (sizeof(T) >= 4) {
        compare_block_butterfly<T, W, IS_EQ>(in, (T)a.constant, c, keep);
    } else {
        // 144-byte stride between the mask images of a wavefront's 8 blocks: at 128 bytes (= all 32 banks) the same byte
        // of every block falls into the same bank and each ds_write is an 8-way conflict (profiles/r03_pmc_sq_counters.csv:
        // SQ_LDS_BANK_CONFLICT = 0.71-0.74 of SQ_LDS_IDX_ACTIVE before); 36 dwords shift each block by 4 banks
        constexpr unsigned MASK_STRIDE = 144;
        __shared__ __attribute__((aligned(16))) char lds[BLOCKS_PER_WG * MASK_STRIDE];
        compare_block_lds<T, W, IS_EQ>(in, (T)a.constant, c, lds + (tid >> 3) * MASK_STRIDE, keep);
    }
    const uint32_t flip = a.invert ? ~0u : 0u;
    const u32x4 out = {keep[0] ^ flip, keep[1] ^ flip, keep[2] ^ flip, keep[3] ^ flip};
    // The mask is 1/8 .. 1/56 of the bytes moved and still what bounds this kernel: with the store suppressed, or aimed at one
    // L2-resident 4 KiB, the same loads run at 7.8-8.5 TB/s (counting the mask), with it at 5.8-6.0 -- a thin write stream inside a
    // read stream costs the DRAM about 2.4x its bytes (profiles/abcompare_maskstore_r03.txt).  The streaming store policy of the
    // unpack kernels (nt + sc1) recovers 3-6 % of that at the narrow widths; walking several tiles per workgroup does not.
    // ONE descriptor per wavefront (its 8 blocks' masks are 1 KiB contiguous), built from readfirstlane'd values: a per-block
    // descriptor differs between the 8 lane groups and hipcc wraps the store in a waterfall loop that runs 8 times (~80
    // instructions of the ~430 of the u16 W=3 kernel, round 3; cdna_hip_programming.md T20)
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t wave_first = tile * BLOCKS_PER_WG + wave * 8u;
    const uint64_t left = a.n_blocks - wave_first;                     // >= 1: this lane's block exists
    const __amdgpu_buffer_rsrc_t mask_rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(a.mask) + wave_first * 128u, 0, (unsigned)(left < 8 ? left : 8) * 128u, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(out, mask_rs, ((tid >> 3) & 7u) * 128u + c * 16u, 0, STORE_AUX);
}

// `waves` (0 = whatever the kernel's own cap allows): workgroups per CU = waves per SIMD, enforced by padding the launch's
// dynamic-LDS request (the kernel never touches it) -- occupancy as a launch parameter, as for the wave-per-block kernels
typedef hipError_t (*compare_launch_t)(const CompareArgs&, int waves, hipStream_t);
template <typename T, int W, bool IS_EQ> hipError_t launch_unpack_compare(const CompareArgs& a0, int waves, hipStream_t s)
{
    if (a0.n_blocks == 0) return hipSuccess;
    CompareArgs a = a0;
    const uint64_t n_tiles = (a.n_blocks + BLOCKS_PER_WG - 1) / BLOCKS_PER_WG;
    a.tiles_per_xcd = (n_tiles + 7) / 8;
    a.window_shift = tile_window_shift(WIN_UNPACK_COMPARE, Elem<T>::BITS, BLOCKS_PER_WG);
    constexpr unsigned STATIC_LDS = sizeof(T) >= 4 ? 0u : BLOCKS_PER_WG * 144u;
    unsigned pad = 0;
    if (waves >= 3 && waves < 8) pad = ((160u * 1024u / (unsigned)waves) & ~1023u) - STATIC_LDS;   // < 64 KiB for waves >= 3
    FL_LAUNCH((k_unpack_compare<T, W, IS_EQ>), dim3((unsigned)(a.tiles_per_xcd * 8)), dim3(WG), pad, s, a);
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

typedef hipError_t (*reduce_launch_t)(const ReduceArgs&, hipStream_t);

inline unsigned plan_grid(ReduceArgs& a, WindowOp op, unsigned type_bits)
{
    const uint64_t n_tiles = (a.n_blocks + BLOCKS_PER_WG - 1) / BLOCKS_PER_WG;
    a.tiles_per_xcd = (n_tiles + 7) / 8;
    a.window_shift = tile_window_shift(op, type_bits, BLOCKS_PER_WG);
    return (unsigned)(a.tiles_per_xcd * 8);
}
template <typename T, int W> hipError_t launch_unpack_block_sums(const ReduceArgs& a0, hipStream_t s)
{
    if (a0.n_blocks == 0) return hipSuccess;
    ReduceArgs a = a0;
    const unsigned grid = plan_grid(a, WIN_UNPACK_BLOCK_SUMS, Elem<T>::BITS);
    FL_LAUNCH((k_unpack_block_sums<T, W>), dim3(grid), dim3(WG), 0, s, a);
    return hipGetLastError();
}
template <typename T> hipError_t launch_block_min_max(const ReduceArgs& a0, hipStream_t s)
{
    if (a0.n_blocks == 0) return hipSuccess;
    ReduceArgs a = a0;
    const unsigned grid = plan_grid(a, WIN_BLOCK_MIN_MAX, Elem<T>::BITS);
    FL_LAUNCH((k_block_min_max<T>), dim3(grid), dim3(WG), 0, s, a);
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
