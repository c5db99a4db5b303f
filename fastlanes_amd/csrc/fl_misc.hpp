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

// transpose / untranspose (transpose.rs:9-23) as an in-register regroup on the cell-column
// mapping: see lane_base / load_lane_runs / store_lane_runs in fl_device.hpp.
template <typename T, bool INVERSE>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(1, PackPolicy<T>::MAXW)))
void k_transpose(StreamArgs a)
{
    constexpr int TB = Elem<T>::BITS;
    uint64_t tile;
    if (!tile_of_workgroup(a, tile)) return;
    const unsigned tid = threadIdx.x;
    // readfirstlane: keeps the wave's store descriptor in SGPRs (no waterfall loop around each store)
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u, c = tid & 7u;
    const uint64_t first_blk = tile * BLOCKS_PER_WG + wave * 8;
    if (first_blk >= a.n_blocks) return;                       // whole wavefront past the end
    const uint64_t blk = first_blk + (lane >> 3);
    const bool valid = blk < a.n_blocks;
    Cell<T> rows[TB];
    if constexpr (!INVERSE) {
        // original order -> transposed: out[index(r,l)] = in[lane_base(l) + r]   (transpose.rs:12-14)
        __shared__ __attribute__((aligned(16))) char lds_in[(WG / 64) * RunExchange<T>::WAVE_BYTES];
        using WS = WaveRowStore<T>;
        __shared__ __attribute__((aligned(16))) char lds_out[(WG / 64) * WS::WAVE_LDS];
        static_for<TB>([&](auto R) { rows[decltype(R)::value] = Cell<T>::zero(); });
        if (valid)
            load_lane_runs_lines<T>(lds_in + (tid >> 3) * RunExchange<T>::BLOCK_BYTES, c,
                                    a.in + blk * (uint64_t)Elem<T>::CELLS_PER_BLOCK, rows);
        const WS ws(a.out, first_blk, a.n_blocks, lds_out + wave * WS::WAVE_LDS, lane);
        static_for<WS::GROUPS>([&](auto G) {
            constexpr int grp = decltype(G)::value;
            static_for<8>([&](auto I) { ws.template put<decltype(I)::value>(rows[WS::row_at(8 * grp + decltype(I)::value)]); });
            ws.template flush<grp>();
        });
    } else {
        // transposed -> original order: out[lane_base(l) + r] = in[index(r,l)]   (transpose.rs:19-21)
        if (!valid) return;                                    // per-block exchange only
        const u32x4* src = a.in + blk * (uint64_t)Elem<T>::CELLS_PER_BLOCK + c;
        static_for<TB>([&](auto J) {   // issued in ascending address order
            rows[WaveRowStore<T>::row_at(decltype(J)::value)] = load_cell<T, true>(src + 8 * decltype(J)::value);
        });
        const TileStore<Elem<T>::CELLS_PER_BLOCK * 16> st(a.out, tile, a.n_blocks, tid);
        __shared__ __attribute__((aligned(16))) char lds[(WG / 64) * RunExchange<T>::WAVE_BYTES];
        store_lane_runs_lines<T>(lds + (tid >> 3) * RunExchange<T>::BLOCK_BYTES, c, rows, st);
    }
}

template <typename T, bool INVERSE>
hipError_t launch_transpose(const StreamArgs& a0, hipStream_t s)
{
    if (a0.n_blocks == 0) return hipSuccess;
    StreamArgs a = a0;
    const unsigned grid = plan_grid(a, INVERSE ? WIN_UNTRANSPOSE : WIN_TRANSPOSE, Elem<T>::BITS);
    FL_LAUNCH((k_transpose<T, INVERSE>), dim3(grid), dim3(WG), 0, s, a);
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
    // mixed-width columns: per-block widths[] / byte offsets[] in HBM (nullptr = every block has `width`, back to back)
    const uint8_t* widths;
    const uint64_t* offsets;
    uint64_t packed_bytes;    // size of the packed column (only read when widths != nullptr)
};

// FL_DEVERR_* of include/fastlanes_amd.h (same values as fl_widths.hpp's DEVERR_*)
__device__ __forceinline__ void single_error(const SingleArgs& a, uint32_t bits)
{
    if (a.err_flag) __hip_atomic_fetch_or(a.err_flag, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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
    const uint64_t blk = gi >> 10;
    if (a.widths && blk >= a.n_blocks) {                      // bitpacking.rs:152 (before the width is known)
        out[k] = 0;
        single_error(a, 2u /* FL_DEVERR_INDEX */);
        return;
    }
    const unsigned W = a.widths ? (unsigned)a.widths[blk] : a.width;
    if (W > TB) {                                             // bitpacking.rs:197 unreachable!()
        out[k] = 0;
        single_error(a, 1u /* FL_DEVERR_WIDTH */);
        return;
    }
    if (W == 0) { out[k] = 0; return; }                       // bitpacking.rs:136-139
    if (blk >= a.n_blocks) {                                  // bitpacking.rs:152
        out[k] = 0;
        single_error(a, 2u /* FL_DEVERR_INDEX */);
        return;
    }
    if (a.widths) {                                           // bitpacking.rs:185-186 debug_assert on the packed length
        const uint64_t off = a.offsets[blk];
        const uint32_t e = ((off & (sizeof(T) - 1)) ? 4u : 0u) | ((off > a.packed_bytes || 128ull * W > a.packed_bytes - off) ? 8u : 0u);
        if (e) {
            out[k] = 0;
            single_error(a, e);
            return;
        }
    }
    const unsigned index = (unsigned)gi & 1023u;
    const unsigned lane = index % LANES;                      // bitpacking.rs:210
    const unsigned s = index / 128;                           // bitpacking.rs:226
    const unsigned o = fl_order((index - s * 128 - lane) / 16);   // bitpacking.rs:227-228
    const unsigned row = o * 8 + s;                           // bitpacking.rs:229
    const T* pk = a.widths ? reinterpret_cast<const T*>(static_cast<const char*>(a.packed) + a.offsets[blk])
                           : static_cast<const T*>(a.packed) + blk * (uint64_t)(1024u * W / TB);
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
    FL_LAUNCH((k_unpack_single<T>), dim3((unsigned)((a.n_indices + WG - 1) / WG)), dim3(WG), 0, s, a);
    return hipGetLastError();
}

template <typename T> stream_launch_t transpose_launcher(bool inverse);
template <typename T> hipError_t unpack_single_launch(const SingleArgs& a, hipStream_t s);

}  // namespace fl
