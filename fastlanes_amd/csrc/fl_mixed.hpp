// fl_mixed.hpp -- mixed-width columns: every block has its own width (BASELINE.json
// config 5).  The reference has no multi-block API; its callers loop
// `unchecked_unpack(width_of(block), ..)` over blocks (bitpacking.rs:109-129).  Here the
// blocks are bucketed by width once (fl_mixed_plan: (block, packed offset) entries sorted by
// width, offsets = exclusive prefix sum of 128*W; tiles of <= 32 same-width blocks, the tile
// list sorted by column position) and the whole column is ONE launch: every workgroup reads
// its tile descriptor (wave-uniform) and branches into the per-(T,W) column code -- the
// device-side form of the reference's `match width { #(W => Self::unpack::<W>(..))* }`
// (bitpacking.rs:115-128).
#pragma once
#include "fl_kernels.hpp"

namespace fl {

// One entry per block, in bucket order: where the block lives on both sides.
struct MixedEntry {
    uint64_t blk;          // block index in the column (unpacked side: blk * 1024 elements)
    uint64_t packed_off;   // byte offset of the block in the packed column
};
// One descriptor per tile of <= 32 same-width blocks (read with scalar loads: wave-uniform).
struct MixedTile {
    uint32_t first_entry;
    uint32_t count_width;  // count | width << 8
    uint64_t first_blk;    // = entries[first_entry].blk        (lowest address of the tile)
    uint64_t first_off;    // = entries[first_entry].packed_off
    uint64_t pad_;
};

struct MixedArgs {
    const char* packed;          // packed column base (bytes)
    char* unpacked;              // unpacked column base (bytes)
    const MixedEntry* entries;   // [n_blocks], buckets in width order, ascending blk inside a bucket
    const MixedTile* tiles;      // [n_tiles]
    uint64_t n_tiles;
    uint64_t tiles_per_xcd;
};

// Streaming-store window over the span of one tile's blocks (they are ascending, so the
// first entry is the lowest address).  WINDOW=false: some tile's span does not fit a 32-bit
// buffer offset (pathologically sparse bucket) -> plain non-temporal global stores.
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
__device__ __forceinline__ void unpack_mixed_tile(const MixedArgs& a, const MixedTile& t, const MixedEntry& e, unsigned c)
{
    constexpr bool NTL = UnpackPolicy<T, W>::NT_LOAD;
    constexpr unsigned BLOCK_BYTES = Elem<T>::CELLS_PER_BLOCK * 16;
    Cell<T> in[W ? W : 1];
    const u32x4* pk = reinterpret_cast<const u32x4*>(a.packed + e.packed_off) + c;
    static_for<W>([&](auto Wd) { in[decltype(Wd)::value] = load_cell<T, NTL>(pk + 8 * decltype(Wd)::value); });
    const SpanStore<WINDOW> st(a.unpacked + t.first_blk * BLOCK_BYTES, a.unpacked + e.blk * BLOCK_BYTES + c * 16);
    unpack_rows_by_address<T, W>(in, [&](auto R, const Cell<T>& v) { st.store(16 * Elem<T>::row_cell(decltype(R)::value), v); });
}

template <typename T, int W, bool WINDOW>
__device__ __forceinline__ void pack_mixed_tile(const MixedArgs& a, const MixedTile& t, const MixedEntry& e, unsigned c)
{
    constexpr int TB = Elem<T>::BITS;
    constexpr unsigned BLOCK_BYTES = Elem<T>::CELLS_PER_BLOCK * 16;
    if constexpr (W == 0) return;
    const u32x4* un = reinterpret_cast<const u32x4*>(a.unpacked + e.blk * BLOCK_BYTES) + c;
    Cell<T> rows[TB];
    static_for<TB>([&](auto J) {   // issued in ascending address order
        rows[WaveRowStore<T>::row_at(decltype(J)::value)] = load_cell<T, true>(un + 8 * decltype(J)::value);
    });
    char* pk = const_cast<char*>(a.packed);
    const SpanStore<WINDOW> st(pk + t.first_off, pk + e.packed_off + c * 16);
    pack_rows<T, W>([&](auto R) { return rows[decltype(R)::value]; },
                    [&](auto Wd, const Cell<T>& v) { st.store(128 * decltype(Wd)::value, v); });
}

template <typename T, bool PACK, bool WINDOW, int... Ws>
__device__ __forceinline__ void dispatch_width(unsigned w, const MixedArgs& a, const MixedTile& t, const MixedEntry& e,
                                               unsigned c, std::integer_sequence<int, Ws...>)
{
    // wave-uniform chain of compares: exactly one body runs per workgroup
    (void)((w == (unsigned)Ws
                ? ((PACK ? pack_mixed_tile<T, Ws, WINDOW>(a, t, e, c)
                         : unpack_mixed_tile<T, Ws, WINDOW>(a, t, e, c)), true)
                : false) || ...);
}

template <typename T, bool PACK, bool WINDOW>
__global__ __launch_bounds__(WG)
__attribute__((amdgpu_waves_per_eu(1, PACK ? PackPolicy<T>::MAXW : UnpackPolicy<T, 0>::MAXW)))
void k_mixed(MixedArgs a)
{
    const uint64_t tile = (uint64_t)(blockIdx.x & 7u) * a.tiles_per_xcd + (blockIdx.x >> 3);
    if (tile >= a.n_tiles) return;
    const MixedTile t = a.tiles[tile];                         // wave-uniform (scalar loads)
    const unsigned count = t.count_width & 0xffu, w = t.count_width >> 8;
    const unsigned tid = threadIdx.x;
    const unsigned g = tid >> 3, c = tid & 7u;
    if (g >= count) return;
    const MixedEntry e = a.entries[t.first_entry + g];         // one 16-byte load per thread
    dispatch_width<T, PACK, WINDOW>(w, a, t, e, c, std::make_integer_sequence<int, Elem<T>::BITS + 1>{});
}

typedef hipError_t (*mixed_launch_t)(const MixedArgs&, hipStream_t);

template <typename T, bool PACK, bool WINDOW>
hipError_t launch_mixed(const MixedArgs& a0, hipStream_t s)
{
    if (a0.n_tiles == 0) return hipSuccess;
    MixedArgs a = a0;
    a.tiles_per_xcd = (a.n_tiles + 7) / 8;
    hipLaunchKernelGGL((k_mixed<T, PACK, WINDOW>), dim3((unsigned)(a.tiles_per_xcd * 8)), dim3(WG), 0, s, a);
    return hipGetLastError();
}

// specialised in fl_inst.hip (families 6 and 7, separate TUs so they build in parallel)
template <typename T> mixed_launch_t mixed_unpack_launcher(bool window);
template <typename T> mixed_launch_t mixed_pack_launcher(bool window);

}  // namespace fl
