// fl_misc.hpp -- transpose / untranspose (transpose.rs:9-36) and batched
// unpack_single (bitpacking.rs:132-179) kernels.
#pragma once
#include "fl_kernels.hpp"

namespace fl {

// transpose.rs:29-36:  tau(i) = (i%16)*64 + FL_ORDER[(i/16)%8]*8 + i/128
__host__ __device__ constexpr unsigned tau(unsigned i)
{
    return (i % 16) * 64 + fl_order((i / 16) % 8) * 8 + i / 128;
}
// inverse: j = a*64 + b*8 + r  ->  r*128 + FL_ORDER[b]*16 + a   (FL_ORDER is self-inverse, lib.rs:53-59)
__host__ __device__ constexpr unsigned tau_inv(unsigned j)
{
    return (j % 8) * 128 + fl_order((j / 8) % 8) * 16 + j / 64;
}

// Per-lane view of the permutation.  Along FL-lane l's row order the transposed positions
// index(r,l) map to T CONSECUTIVE original positions:
//     tau(index(r, l)) = lane_base(l) + r,   lane_base(l) = (l%16)*64 + FL_ORDER[l/16]*8
// (verified against transpose.rs:29-36 for every T, r, l by tests/test_oracle_properties.py).
// So the cell-column thread that owns lanes n*c .. n*c+n-1 for all T rows (fl_device.hpp)
// holds, per lane, one contiguous run of T elements = T*sizeof(T) bytes of the original
// order, and the permutation is a pure in-register regroup -- no LDS, no cross-lane traffic:
//   transpose   : per lane, gather the run with 16-byte loads, regroup, write the T
//                 transposed rows as full 128-byte-line stores (TileStore);
//   untranspose : read the T rows as full-line loads, regroup, write each lane's run with
//                 16-byte stores (plain write-back stores so L2 merges a run's pieces).
__host__ __device__ constexpr unsigned lane_base(unsigned l) { return (l % 16) * 64 + fl_order(l / 16) * 8; }

template <typename T> __device__ __forceinline__ uint64_t cell_get(const Cell<T>& c, int e)
{
    if constexpr (sizeof(T) == 8) return c.x[e];
    else if constexpr (sizeof(T) == 4) return c.x[e];
    else if constexpr (sizeof(T) == 2) return (c.x[e / 2] >> (16 * (e % 2))) & 0xffffu;
    else return (c.x[e / 4] >> (8 * (e % 4))) & 0xffu;
}
template <typename T> __device__ __forceinline__ void cell_or(Cell<T>& c, int e, uint64_t v)
{
    if constexpr (sizeof(T) == 8) c.x[e] = v;
    else if constexpr (sizeof(T) == 4) c.x[e] = (uint32_t)v;
    else if constexpr (sizeof(T) == 2) c.x[e / 2] |= (uint32_t)v << (16 * (e % 2));
    else c.x[e / 4] |= (uint32_t)v << (8 * (e % 4));
}

template <typename T, bool INVERSE>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(1, PackPolicy<T>::MAXW)))
void k_transpose(StreamArgs a)
{
    constexpr int TB = Elem<T>::BITS;
    constexpr int N = Elem<T>::PER_CELL;                 // FL lanes per thread
    constexpr int E = sizeof(T);
    constexpr int RUN_BYTES = TB * E;                    // one lane's run: 8 B (u8) .. 512 B (u64)
    constexpr int PIECE = RUN_BYTES < 16 ? RUN_BYTES : 16;
    constexpr int PIECES = RUN_BYTES / PIECE;
    constexpr int PER_PIECE = PIECE / E;                 // elements per piece
    typedef uint32_t piece_t __attribute__((ext_vector_type(PIECE / 4)));
    uint64_t tile;
    if (!tile_of_workgroup(a, tile)) return;
    const unsigned tid = threadIdx.x;
    const uint64_t blk = tile * BLOCKS_PER_WG + (tid >> 3);
    const unsigned c = tid & 7u;
    if (blk >= a.n_blocks) return;
    const char* in_blk = reinterpret_cast<const char*>(a.in) + blk * (uint64_t)(1024 * E);
    char* out_blk = reinterpret_cast<char*>(a.out) + blk * (uint64_t)(1024 * E);
    Cell<T> rows[TB];

    if constexpr (!INVERSE) {
        // original order -> transposed: out[index(r,l)] = in[lane_base(l) + r]   (transpose.rs:12-14)
        static_for<TB>([&](auto R) { rows[decltype(R)::value] = Cell<T>::zero(); });
        static_for<N>([&](auto EE) {
            constexpr int e = decltype(EE)::value;
            const char* run = in_blk + (uint64_t)lane_base(N * c + e) * E;
            piece_t p[PIECES];
            static_for<PIECES>([&](auto K) { p[decltype(K)::value] = *reinterpret_cast<const piece_t*>(run + PIECE * decltype(K)::value); });
            static_for<TB>([&](auto R) {
                constexpr int r = decltype(R)::value;
                constexpr int k = r / PER_PIECE, j = r % PER_PIECE;
                uint64_t v;
                if constexpr (E == 8) v = (uint64_t)p[k][2 * j] | ((uint64_t)p[k][2 * j + 1] << 32);
                else if constexpr (E == 4) v = p[k][j];
                else if constexpr (E == 2) v = (p[k][j / 2] >> (16 * (j % 2))) & 0xffffu;
                else v = (p[k][j / 4] >> (8 * (j % 4))) & 0xffu;
                cell_or<T>(rows[r], e, v);
            });
        });
        const TileStore<Elem<T>::CELLS_PER_BLOCK * 16> st(a.out, tile, a.n_blocks, tid);
        static_for<TB>([&](auto R) { st.store(Elem<T>::row_cell(decltype(R)::value), rows[decltype(R)::value]); });
    } else {
        // transposed -> original order: out[lane_base(l) + r] = in[index(r,l)]   (transpose.rs:19-21)
        const u32x4* src = reinterpret_cast<const u32x4*>(in_blk) + c;
        static_for<TB>([&](auto R) {
            rows[decltype(R)::value] = load_cell<T, true>(src + Elem<T>::row_cell(decltype(R)::value));
        });
        static_for<N>([&](auto EE) {
            constexpr int e = decltype(EE)::value;
            char* run = out_blk + (uint64_t)lane_base(N * c + e) * E;
            static_for<PIECES>([&](auto K) {
                constexpr int k = decltype(K)::value;
                piece_t p;
                static_for<PIECE / 4>([&](auto D) {
                    constexpr int d = decltype(D)::value;
                    uint32_t w = 0;
                    if constexpr (E == 8) {
                        const uint64_t v = cell_get<T>(rows[k * PER_PIECE + d / 2], e);
                        w = (uint32_t)(v >> (32 * (d % 2)));
                    } else {
                        static_for<4 / (E < 4 ? E : 4)>([&](auto J) {
                            constexpr int j = decltype(J)::value;
                            w |= (uint32_t)cell_get<T>(rows[k * PER_PIECE + d * (4 / E) + j], e) << (8 * E * j);
                        });
                    }
                    p[d] = w;
                });
                *reinterpret_cast<piece_t*>(run + PIECE * k) = p;
            });
        });
    }
}

template <typename T, bool INVERSE>
hipError_t launch_transpose(const StreamArgs& a0, hipStream_t s)
{
    if (a0.n_blocks == 0) return hipSuccess;
    StreamArgs a = a0;
    const unsigned grid = plan_grid(a);
    hipLaunchKernelGGL((k_transpose<T, INVERSE>), dim3(grid), dim3(WG), 0, s, a);
    return hipGetLastError();
}

struct SingleArgs {
    const void* packed;
    const uint64_t* indices;
    void* out;
    uint32_t* err_flag;
    uint64_t n_blocks;
    uint64_t n_indices;
    unsigned width;
};

// bitpacking.rs:132-179 with the lookup tables of :207-232 computed in closed form.
template <typename T>
__global__ __launch_bounds__(WG) void k_unpack_single(SingleArgs a)
{
    constexpr unsigned TB = Elem<T>::BITS;
    constexpr unsigned LANES = Elem<T>::LANES;
    const uint64_t k = (uint64_t)blockIdx.x * WG + threadIdx.x;
    if (k >= a.n_indices) return;
    const uint64_t gi = a.indices[k];
    T* out = static_cast<T*>(a.out);
    const unsigned W = a.width;
    if (W == 0) { out[k] = 0; return; }                       // bitpacking.rs:136-139
    const uint64_t blk = gi >> 10;
    if (blk >= a.n_blocks) {                                  // bitpacking.rs:152
        out[k] = 0;
        if (a.err_flag) *a.err_flag = 1u;
        return;
    }
    const unsigned index = (unsigned)gi & 1023u;
    const unsigned lane = index % LANES;                      // bitpacking.rs:210
    const unsigned s = index / 128;                           // bitpacking.rs:226
    const unsigned o = fl_order((index - s * 128 - lane) / 16);   // bitpacking.rs:227-228
    const unsigned row = o * 8 + s;                           // bitpacking.rs:229
    const T* pk = static_cast<const T*>(a.packed) + blk * (uint64_t)(1024u * W / TB);
    if (W == TB) { out[k] = pk[LANES * row + lane]; return; } // bitpacking.rs:159-162
    const T mask = (T)(((T)1 << W) - (T)1);
    const unsigned start_bit = row * W;
    const unsigned start_word = start_bit / TB;
    const unsigned lo_shift = start_bit % TB;
    const unsigned remaining = TB - lo_shift;
    T v = (T)(pk[LANES * start_word + lane] >> lo_shift);
    if (remaining < W) v = (T)(v | (T)(pk[LANES * (start_word + 1) + lane] << remaining));
    out[k] = (T)(v & mask);
}

template <typename T>
hipError_t launch_unpack_single(const SingleArgs& a, hipStream_t s)
{
    if (a.n_indices == 0) return hipSuccess;
    hipLaunchKernelGGL((k_unpack_single<T>), dim3((unsigned)((a.n_indices + WG - 1) / WG)), dim3(WG), 0, s, a);
    return hipGetLastError();
}

template <typename T> stream_launch_t transpose_launcher(bool inverse);
template <typename T> hipError_t unpack_single_launch(const SingleArgs& a, hipStream_t s);

}  // namespace fl
