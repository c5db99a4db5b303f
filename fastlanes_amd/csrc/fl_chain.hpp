// fl_chain.hpp -- Delta's stateful bodies on the wave-per-block mapping (fl_widths.hpp):
//   CHAIN_UNDELTA_PACK  Delta::undelta_pack::<W>  (delta.rs:47-63)   packed W rows + base  -> T rows
//   CHAIN_UNDELTA       Delta::undelta            (delta.rs:36-45)   T rows + base         -> T rows
//   CHAIN_DELTA         Delta::delta              (delta.rs:24-33)   T rows + base         -> T rows
// One wavefront per block, every global access 1 KiB contiguous.  The per-FL-lane chain over the T logical rows
// (row order matters: macros.rs:119, delta.rs:56-61) is cut into 8 segments of R = T/8 CONSECUTIVE rows: lane
// (i = lane/8, c = lane%8) owns rows R*i .. R*i+R-1 of cell column c.  It takes their values from the wave's LDS
// image of the input (funnel-shifted out of the packed rows, or read at row_cell(r)), runs the chain locally, and
// the 8 segments are stitched with a 3-step exclusive scan of the segment totals across lane groups (undelta) or by
// reading the previous segment's last row (delta).  The results go back into the LDS image at their row_cell
// positions and leave in address order.  LDS is wave-local: no s_barrier.
#pragma once
#include "fl_widths.hpp"

namespace fl {

enum ChainMode { CHAIN_UNDELTA_PACK = 0, CHAIN_UNDELTA = 1, CHAIN_DELTA = 2 };

struct ChainArgs {
    const char* in;          // packed column (UNDELTA_PACK) or unpacked column
    char* out;               // unpacked column
    const char* bases;       // [n_blocks][128 bytes]
    uint64_t n_blocks;
    uint64_t tiles_per_xcd;
    unsigned width;          // UNDELTA_PACK only
};

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

template <typename T, int MODE>
__global__ __launch_bounds__(WG) void k_chain(ChainArgs a)
{
    using G = WaveBlock<T>;
    constexpr int TB = G::TB;
    constexpr int R = TB / 8;                                  // consecutive logical rows per lane
    extern __shared__ __attribute__((aligned(16))) char lds_all[];
    const uint64_t n_tiles = (a.n_blocks + (WG / 64) - 1) / (WG / 64);
    const uint64_t tile = (uint64_t)(blockIdx.x & 7u) * a.tiles_per_xcd + (blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    const unsigned tid = threadIdx.x;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
    const uint64_t blk = tile * (WG / 64) + wave;
    if (blk >= a.n_blocks) return;
    char* lds = lds_all + wave * G::BLOCK_BYTES;
    const unsigned i = lane >> 3, c16 = (lane & 7u) * 16u;
    const unsigned w = MODE == CHAIN_UNDELTA_PACK ? a.width : (unsigned)TB;

    // ---- input block -> LDS image (1 KiB-contiguous loads) ---------------------------------------------------
    const unsigned in_bytes = MODE == CHAIN_UNDELTA_PACK ? 128u * w : G::BLOCK_BYTES;
    const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(a.in) + blk * (uint64_t)in_bytes, 0, in_bytes, 0x00020000);
    u32x4 img[G::GROUPS];
    static_for<G::GROUPS>([&](auto Gi) {
        constexpr int g = decltype(Gi)::value;
        if (8u * g < w) img[g] = __builtin_amdgcn_raw_buffer_load_b128(in_rs, lane * 16u + g * 1024u, 0, MODE == CHAIN_UNDELTA_PACK ? 0 : 2);
    });
    // base[lane] of this cell column (delta.rs:26,38,56): one 16-byte cell per column, behind the data loads
    const Cell<T> base = __builtin_bit_cast(Cell<T>, *reinterpret_cast<const u32x4*>(a.bases + blk * 128u + c16 + opaque_zero()));
    static_for<G::GROUPS>([&](auto Gi) {
        constexpr int g = decltype(Gi)::value;
        if (8u * g < w) *reinterpret_cast<u32x4*>(lds + lane * 16u + g * 1024u) = img[g];
    });
    wave_lds_fence();

    // ---- this lane's R consecutive rows ----------------------------------------------------------------------
    Cell<T> x[R];
    const unsigned r0 = R * i;
    if constexpr (MODE == CHAIN_UNDELTA_PACK) {
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
    } else {
        static_for<R>([&](auto J) {
            x[decltype(J)::value] = __builtin_bit_cast(Cell<T>, *reinterpret_cast<const u32x4*>(lds + G::row_cell_rt(r0 + decltype(J)::value) * 16u + c16));
        });
    }
    if constexpr (MODE == CHAIN_DELTA) {
        // out[idx] = in[idx] - prev; prev = in[idx]  (delta.rs:28-30): the row before the segment is the previous
        // lane group's last row (still in the LDS image), or base for the first segment
        Cell<T> prev = base;
        if (i != 0) prev = __builtin_bit_cast(Cell<T>, *reinterpret_cast<const u32x4*>(lds + G::row_cell_rt(r0 - 1u) * 16u + c16));
        wave_lds_fence();                                      // every lane has read its neighbour's row before anyone overwrites
        static_for<R>([&](auto J) {
            const Cell<T> cur = x[decltype(J)::value];
            x[decltype(J)::value] = cur.sub(prev);
            prev = cur;
        });
    } else {
        // next = elem + prev; out[idx] = next; prev = next  (delta.rs:40-42,58-60): local running sum, then the
        // exclusive scan of the segment totals over the 8 lane groups (Hillis-Steele, 3 steps), base entering at segment 0
        if (i == 0) x[0] = x[0].add(base);
        static_for<R - 1>([&](auto J) { x[decltype(J)::value + 1] = x[decltype(J)::value + 1].add(x[decltype(J)::value]); });
        Cell<T> incl = x[R - 1];
        static_for<3>([&](auto S) {
            constexpr unsigned d = 1u << decltype(S)::value;
            incl = incl.add(cell_from_group_below<T>(incl, lane, d));
        });
        const Cell<T> excl = cell_from_group_below<T>(incl, lane, 1);
        static_for<R>([&](auto J) { x[decltype(J)::value] = x[decltype(J)::value].add(excl); });
        if constexpr (MODE == CHAIN_UNDELTA_PACK) wave_lds_fence();   // the packed image is dead only once every lane has unpacked
    }

    // ---- results -> LDS image at their address-order cells -> 1 KiB-contiguous stores ------------------------
    static_for<R>([&](auto J) {
        *reinterpret_cast<u32x4*>(lds + G::row_cell_rt(r0 + decltype(J)::value) * 16u + c16) = __builtin_bit_cast(u32x4, x[decltype(J)::value]);
    });
    wave_lds_fence();
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(a.out + blk * G::BLOCK_BYTES, 0, G::BLOCK_BYTES, 0x00020000);
    static_for<G::GROUPS>([&](auto K) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(lds + lane * 16u + decltype(K)::value * 1024u);
        __builtin_amdgcn_raw_buffer_store_b128(v, out_rs, lane * 16u + decltype(K)::value * 1024u, 0, STORE_AUX);
    });
}

typedef hipError_t (*chain_launch_t)(const ChainArgs&, int waves, hipStream_t);

template <typename T, int MODE>
hipError_t launch_chain(const ChainArgs& a0, int waves, hipStream_t s)
{
    if (a0.n_blocks == 0) return hipSuccess;
    ChainArgs a = a0;
    const uint64_t n_tiles = (a.n_blocks + (WG / 64) - 1) / (WG / 64);
    a.tiles_per_xcd = (n_tiles + 7) / 8;
    if (a.tiles_per_xcd * 8 > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_chain<T, MODE>), dim3((unsigned)(a.tiles_per_xcd * 8)), dim3(WG), widths_lds_bytes<T>(waves), s, a);
    return hipGetLastError();
}

template <typename T> chain_launch_t chain_launcher(int mode);

}  // namespace fl
