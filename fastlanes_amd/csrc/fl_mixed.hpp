// fl_mixed.hpp -- mixed-width columns: every block has its own width (BASELINE.json
// config 5).  The reference has no multi-block API; its callers loop
// `unchecked_unpack(width_of(block), ..)` over blocks (bitpacking.rs:109-129).  Here the
// blocks are bucketed by width once (fl_mixed_plan: ids sorted by width, byte offsets =
// exclusive prefix sum of 128*W) and each bucket is one launch of the same per-(T,W)
// column kernels, reading/writing blocks through the id/offset indirection, so every
// launch stays wave-uniform in W.
#pragma once
#include "fl_kernels.hpp"

namespace fl {

struct MixedArgs {
    const char* packed;        // packed column base (bytes)
    char* unpacked;            // unpacked column base (bytes)
    const uint32_t* ids;       // this bucket's block ids, ascending
    const uint64_t* offsets;   // byte offset of every block of the column in `packed`
    uint64_t m;                // blocks in this bucket
    uint64_t tiles_per_xcd;
};

__device__ __forceinline__ bool tile_of_workgroup(const MixedArgs& a, uint64_t& tile)
{
    const uint64_t n_tiles = (a.m + BLOCKS_PER_WG - 1) / BLOCKS_PER_WG;
    tile = (uint64_t)(blockIdx.x & 7u) * a.tiles_per_xcd + (blockIdx.x >> 3);
    return tile < n_tiles;
}

// Streaming-store window over the span of one tile's blocks (they are ascending, so the
// first entry is the lowest address).  WINDOW=false: the span does not fit a 32-bit buffer
// offset (pathologically sparse bucket) -> plain non-temporal global stores.
template <bool WINDOW> struct SpanStore {
    __amdgpu_buffer_rsrc_t rs;
    char* base;
    uint64_t delta;
    __device__ __forceinline__ SpanStore(char* first, char* mine)
    {
        base = mine;
        delta = (uint64_t)(mine - first);
        if constexpr (WINDOW) rs = __builtin_amdgcn_make_buffer_rsrc(first, 0, 0xFFFFFFFFu, 0x00020000);
    }
    template <typename T> __device__ __forceinline__ void store(unsigned byte_off, const Cell<T>& v) const
    {
        if constexpr (WINDOW)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (unsigned)delta + byte_off, 0, STORE_AUX);
        else
            __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(base + byte_off));
    }
};

template <typename T, int W, bool WINDOW>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(1, UnpackPolicy<T, W>::MAXW)))
void k_unpack_mixed(MixedArgs a)
{
    constexpr bool NTL = UnpackPolicy<T, W>::NT_LOAD;
    constexpr unsigned BLOCK_BYTES = Elem<T>::CELLS_PER_BLOCK * 16;
    uint64_t tile;
    if (!tile_of_workgroup(a, tile)) return;
    const unsigned tid = threadIdx.x;
    const uint64_t e = tile * BLOCKS_PER_WG + (tid >> 3);
    const unsigned c = tid & 7u;
    if (e >= a.m) return;
    const uint64_t first = a.ids[tile * BLOCKS_PER_WG];   // wave-uniform
    const uint64_t blk = a.ids[e];
    Cell<T> in[W ? W : 1];
    const u32x4* pk = reinterpret_cast<const u32x4*>(a.packed + a.offsets[blk]) + c;
    static_for<W>([&](auto Wd) { in[decltype(Wd)::value] = load_cell<T, NTL>(pk + 8 * decltype(Wd)::value); });
    const SpanStore<WINDOW> st(a.unpacked + first * BLOCK_BYTES, a.unpacked + blk * BLOCK_BYTES + c * 16);
    unpack_rows<T, W>(in, [&](auto R, const Cell<T>& v) { st.store(16 * Elem<T>::row_cell(decltype(R)::value), v); });
}

template <typename T, int W, bool WINDOW>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(1, PackPolicy<T>::MAXW)))
void k_pack_mixed(MixedArgs a)
{
    constexpr int TB = Elem<T>::BITS;
    constexpr unsigned BLOCK_BYTES = Elem<T>::CELLS_PER_BLOCK * 16;
    uint64_t tile;
    if (!tile_of_workgroup(a, tile)) return;
    const unsigned tid = threadIdx.x;
    const uint64_t e = tile * BLOCKS_PER_WG + (tid >> 3);
    const unsigned c = tid & 7u;
    if (e >= a.m) return;
    if constexpr (W == 0) return;
    const uint64_t first = a.ids[tile * BLOCKS_PER_WG];
    const uint64_t blk = a.ids[e];
    const u32x4* un = reinterpret_cast<const u32x4*>(a.unpacked + blk * BLOCK_BYTES) + c;
    Cell<T> rows[TB];
    static_for<TB>([&](auto R) {
        rows[decltype(R)::value] = load_cell<T, true>(un + Elem<T>::row_cell(decltype(R)::value));
    });
    char* pk = const_cast<char*>(a.packed);
    const SpanStore<WINDOW> st(pk + a.offsets[first], pk + a.offsets[blk] + c * 16);
    pack_rows<T, W>([&](auto R) { return rows[decltype(R)::value]; },
                    [&](auto Wd, const Cell<T>& v) { st.store(128 * decltype(Wd)::value, v); });
}

typedef hipError_t (*mixed_launch_t)(const MixedArgs&, hipStream_t);

template <typename T, int W, bool PACK, bool WINDOW>
hipError_t launch_mixed(const MixedArgs& a0, hipStream_t s)
{
    if (a0.m == 0 || (PACK && W == 0)) return hipSuccess;
    MixedArgs a = a0;
    const uint64_t n_tiles = (a.m + BLOCKS_PER_WG - 1) / BLOCKS_PER_WG;
    a.tiles_per_xcd = (n_tiles + 7) / 8;
    const unsigned grid = (unsigned)(a.tiles_per_xcd * 8);
    if constexpr (PACK) hipLaunchKernelGGL((k_pack_mixed<T, W, WINDOW>), dim3(grid), dim3(WG), 0, s, a);
    else hipLaunchKernelGGL((k_unpack_mixed<T, W, WINDOW>), dim3(grid), dim3(WG), 0, s, a);
    return hipGetLastError();
}

// [width][window?]
template <typename T> struct MixedTable { mixed_launch_t fn[Elem<T>::BITS + 1][2]; };
template <typename T, bool PACK, int... Ws>
constexpr MixedTable<T> make_mixed_table(std::integer_sequence<int, Ws...>)
{
    return MixedTable<T>{{{&launch_mixed<T, Ws, PACK, false>, &launch_mixed<T, Ws, PACK, true>}...}};
}
template <typename T, bool PACK> const MixedTable<T>& mixed_table_impl();

}  // namespace fl
