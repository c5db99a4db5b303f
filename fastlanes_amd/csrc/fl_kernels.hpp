// fl_kernels.hpp -- batched gfx950 kernels over n_blocks contiguous 1024-value
// blocks, plus the runtime-width launch tables (the analogue of the reference's
// `match width { #(W => Self::unpack::<W>(..))* }`, bitpacking.rs:82-95,115-128).
//
// Thread mapping: global thread g -> block g/8, cell column g%8 (fl_device.hpp).
// Every kernel is a pure stream: each input byte is read once, each output byte
// written once, 16 B per lane per access; no LDS, no cross-lane traffic, no
// inter-workgroup communication.
#pragma once
#include "fl_device.hpp"

namespace fl {

constexpr int WG = 256;             // 4 wavefronts; 32 blocks per workgroup
constexpr int BLOCKS_PER_WG = WG / 8;

enum UnpackBody { BODY_STORE = 0, BODY_ADD_REF = 1, BODY_UNDELTA = 2 };

// Kernel argument block shared by all streaming kernels.
struct StreamArgs {
    const u32x4* in;       // packed (unpack family) or unpacked (pack family)
    u32x4* out;
    const void* aux;       // references [n_blocks*aux_stride] or bases [n_blocks][LANES]
    uint64_t aux_stride;   // FoR: 0 = one scalar for all blocks, 1 = one per block
    uint64_t n_blocks;
};

// unpack / unfor_pack / undelta_pack  (bitpacking.rs:98-107, ffor.rs:38-50,
// delta.rs:47-63): packed W cell-rows -> T cell-rows.
template <typename T, int W, int BODY, bool NT>
__global__ __launch_bounds__(WG) void k_unpack(StreamArgs a)
{
    constexpr int TB = Elem<T>::BITS;
    const uint64_t g = (uint64_t)blockIdx.x * WG + threadIdx.x;
    const uint64_t blk = g >> 3;
    const unsigned c = (unsigned)g & 7u;
    if (blk >= a.n_blocks) return;

    Cell<T> in[W ? W : 1];
    const u32x4* pk = a.in + blk * (uint64_t)(8 * W) + c;
    static_for<W>([&](auto Wd) { in[decltype(Wd)::value] = load_cell<T, NT>(pk + 8 * decltype(Wd)::value); });

    u32x4* un = a.out + blk * (uint64_t)Elem<T>::CELLS_PER_BLOCK + c;
    if constexpr (BODY == BODY_STORE) {
        unpack_rows<T, W>(in, [&](auto R, const Cell<T>& v) {
            store_cell<T, NT>(un + Elem<T>::row_cell(decltype(R)::value), v);   // bitpacking.rs:103-105
        });
    } else if constexpr (BODY == BODY_ADD_REF) {
        const T* refs = static_cast<const T*>(a.aux);
        const Cell<T> ref = Cell<T>::splat(refs[blk * a.aux_stride]);
        unpack_rows<T, W>(in, [&](auto R, const Cell<T>& v) {
            store_cell<T, NT>(un + Elem<T>::row_cell(decltype(R)::value), v.add(ref));   // ffor.rs:46-48
        });
    } else {
        // base[lane] for this column's lanes = cell c of the block's 128-byte base row
        const u32x4* bases = static_cast<const u32x4*>(a.aux);
        Cell<T> prev = load_cell<T, NT>(bases + blk * 8 + c);                  // delta.rs:56
        unpack_rows<T, W>(in, [&](auto R, const Cell<T>& v) {
            prev = v.add(prev);                                               // delta.rs:58-60
            store_cell<T, NT>(un + Elem<T>::row_cell(decltype(R)::value), prev);
        });
    }
}

// pack / for_pack  (bitpacking.rs:65-74, ffor.rs:24-36): T cell-rows -> W cell-rows.
template <typename T, int W, bool FOR, bool NT>
__global__ __launch_bounds__(WG) void k_pack(StreamArgs a)
{
    constexpr int TB = Elem<T>::BITS;
    const uint64_t g = (uint64_t)blockIdx.x * WG + threadIdx.x;
    const uint64_t blk = g >> 3;
    const unsigned c = (unsigned)g & 7u;
    if (blk >= a.n_blocks) return;
    if constexpr (W == 0) return;                                             // macros.rs:52-53

    const u32x4* un = a.in + blk * (uint64_t)Elem<T>::CELLS_PER_BLOCK + c;
    u32x4* pk = a.out + blk * (uint64_t)(8 * W) + c;
    Cell<T> ref = Cell<T>::zero();
    if constexpr (FOR) ref = Cell<T>::splat(static_cast<const T*>(a.aux)[blk * a.aux_stride]);

    // Issue all T row loads up front (they are independent), then combine.
    Cell<T> rows[TB];
    static_for<TB>([&](auto R) {
        rows[decltype(R)::value] = load_cell<T, NT>(un + Elem<T>::row_cell(decltype(R)::value));
    });
    pack_rows<T, W>(
        [&](auto R) {
            if constexpr (FOR) return rows[decltype(R)::value].sub(ref);      // ffor.rs:32-34
            else return rows[decltype(R)::value];                             // bitpacking.rs:70-72
        },
        [&](auto Wd, const Cell<T>& v) { store_cell<T, NT>(pk + 8 * decltype(Wd)::value, v); });
}

// delta / undelta  (delta.rs:24-45): T cell-rows -> T cell-rows, per-lane chain.
template <typename T, bool INVERSE, bool NT>
__global__ __launch_bounds__(WG) void k_delta(StreamArgs a)
{
    constexpr int TB = Elem<T>::BITS;
    const uint64_t g = (uint64_t)blockIdx.x * WG + threadIdx.x;
    const uint64_t blk = g >> 3;
    const unsigned c = (unsigned)g & 7u;
    if (blk >= a.n_blocks) return;
    const u32x4* src = a.in + blk * (uint64_t)Elem<T>::CELLS_PER_BLOCK + c;
    u32x4* dst = a.out + blk * (uint64_t)Elem<T>::CELLS_PER_BLOCK + c;
    Cell<T> prev = load_cell<T, NT>(static_cast<const u32x4*>(a.aux) + blk * 8 + c);
    Cell<T> rows[TB];
    static_for<TB>([&](auto R) {
        rows[decltype(R)::value] = load_cell<T, NT>(src + Elem<T>::row_cell(decltype(R)::value));
    });
    static_for<TB>([&](auto R) {
        constexpr int row = decltype(R)::value;
        if constexpr (INVERSE) {
            prev = rows[row].add(prev);                                       // delta.rs:40-42
            store_cell<T, NT>(dst + Elem<T>::row_cell(row), prev);
        } else {
            store_cell<T, NT>(dst + Elem<T>::row_cell(row), rows[row].sub(prev));   // delta.rs:28-30
            prev = rows[row];
        }
    });
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
typedef hipError_t (*stream_launch_t)(const StreamArgs&, hipStream_t);

inline unsigned grid_for(uint64_t n_blocks) { return (unsigned)((n_blocks + BLOCKS_PER_WG - 1) / BLOCKS_PER_WG); }

template <typename T, int W, int BODY, bool NT>
hipError_t launch_unpack(const StreamArgs& a, hipStream_t s)
{
    if (a.n_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL((k_unpack<T, W, BODY, NT>), dim3(grid_for(a.n_blocks)), dim3(WG), 0, s, a);
    return hipGetLastError();
}
template <typename T, int W, bool FOR, bool NT>
hipError_t launch_pack(const StreamArgs& a, hipStream_t s)
{
    if (a.n_blocks == 0 || W == 0) return hipSuccess;
    hipLaunchKernelGGL((k_pack<T, W, FOR, NT>), dim3(grid_for(a.n_blocks)), dim3(WG), 0, s, a);
    return hipGetLastError();
}
template <typename T, bool INVERSE, bool NT>
hipError_t launch_delta(const StreamArgs& a, hipStream_t s)
{
    if (a.n_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL((k_delta<T, INVERSE, NT>), dim3(grid_for(a.n_blocks)), dim3(WG), 0, s, a);
    return hipGetLastError();
}

// Runtime width -> instance table, index 0..T inclusive.
template <typename T> struct WidthTable { stream_launch_t fn[Elem<T>::BITS + 1]; };

template <typename T, int BODY, bool NT, int... Ws>
constexpr WidthTable<T> make_unpack_table(std::integer_sequence<int, Ws...>)
{
    return WidthTable<T>{{&launch_unpack<T, Ws, BODY, NT>...}};
}
template <typename T, bool FOR, bool NT, int... Ws>
constexpr WidthTable<T> make_pack_table(std::integer_sequence<int, Ws...>)
{
    return WidthTable<T>{{&launch_pack<T, Ws, FOR, NT>...}};
}

// Specialised once per (element type, family) in fl_inst.hip
template <typename T, int BODY> const WidthTable<T>& unpack_table_impl();
template <typename T, bool FOR> const WidthTable<T>& pack_table_impl();
template <typename T> stream_launch_t delta_launcher(bool inverse);

}  // namespace fl
