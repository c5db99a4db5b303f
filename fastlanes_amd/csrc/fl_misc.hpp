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

// bitpacking.rs:132-179 with the lookup tables of :207-232 computed in closed form.  A lookup in three steps, so that a thread can
// take SEVERAL lookups through each step together (their loads overlap instead of queueing behind one another):
//   single_locate   the index -> block, width, checks (bitpacking.rs:136-139,152,185-186,197) -> where its one or two words live
//   (the loads)     word 0 and, for a straddling field, word 1 (the same word again otherwise: one more hit on the same line)
//   single_extract  shift / or / mask (bitpacking.rs:164-178)
template <typename T> struct SingleWhere {
    const T* w0;             // nullptr: the result is `value` (W == 0, or a failed check: 0)
    const T* w1;
    unsigned lo_shift, width;
    uint32_t err;            // FL_DEVERR_* bits of this lookup
};
template <typename T>
__device__ __forceinline__ SingleWhere<T> single_locate(const SingleArgs& a, uint64_t gi, unsigned W, uint64_t off, bool have_meta)
{
    constexpr unsigned TB = Elem<T>::BITS;
    constexpr unsigned LANES = Elem<T>::LANES;
    SingleWhere<T> r{nullptr, nullptr, 0u, W, 0u};
    const uint64_t blk = gi >> 10;
    if (blk >= a.n_blocks) { r.err = W > TB && !a.widths ? 1u : 2u; if (W == 0 && !a.widths) r.err = 0u; return r; }   // see k_unpack_single's order of checks
    if (W > TB) { r.err = 1u /* FL_DEVERR_WIDTH */; return r; }            // bitpacking.rs:197 unreachable!()
    if (W == 0) return r;                                                     // bitpacking.rs:136-139
    if (have_meta) {                                                          // bitpacking.rs:185-186 debug_assert on the packed length
        const uint32_t e = ((off & (sizeof(T) - 1)) ? 4u : 0u) | ((off > a.packed_bytes || 128ull * W > a.packed_bytes - off) ? 8u : 0u);
        if (e) { r.err = e; return r; }
    }
    const unsigned index = (unsigned)gi & 1023u;
    const unsigned lane = index % LANES;                      // bitpacking.rs:210
    const unsigned s = index / 128;                           // bitpacking.rs:226
    const unsigned o = fl_order((index - s * 128 - lane) / 16);   // bitpacking.rs:227-228
    const unsigned row = o * 8 + s;                           // bitpacking.rs:229
    const T* pk = have_meta ? reinterpret_cast<const T*>(static_cast<const char*>(a.packed) + off)
                            : static_cast<const T*>(a.packed) + blk * (uint64_t)(1024u * W / TB);
    const unsigned start_bit = row * W;
    const unsigned start_word = start_bit / TB;
    r.lo_shift = start_bit % TB;
    r.w0 = pk + LANES * start_word + lane;
    r.w1 = (W < TB && TB - r.lo_shift < W) ? r.w0 + LANES : r.w0;            // bitpacking.rs:168-172: the field straddles two words
    return r;
}
template <typename T> __device__ __forceinline__ T single_extract(const SingleWhere<T>& r, T v0, T v1)
{
    constexpr unsigned TB = Elem<T>::BITS;
    if (!r.w0) return (T)0;
    if (r.width == TB) return v0;                             // bitpacking.rs:159-162
    const T mask = (T)(((T)1 << r.width) - (T)1);
    T v = (T)(v0 >> r.lo_shift);
    if (r.w1 != r.w0) v = (T)(v | (T)(v1 << (TB - r.lo_shift)));
    return (T)(v & mask);
}

// one lookup per thread: any alignment, any count (the tail of the vector form below)
template <typename T>
__global__ __launch_bounds__(WG) void k_unpack_single(SingleArgs a, uint64_t first)
{
    const uint64_t k = first + (uint64_t)blockIdx.x * WG + threadIdx.x;
    if (k >= a.n_indices) return;
    const uint64_t gi = a.indices[k];
    T* out = static_cast<T*>(a.out);
    const uint64_t blk = gi >> 10;
    unsigned W = a.width;
    uint64_t off = 0;
    if (a.widths) {
        if (blk >= a.n_blocks) {                              // bitpacking.rs:152 (before the width is known)
            out[k] = 0;
            single_error(a, 2u /* FL_DEVERR_INDEX */);
            return;
        }
        W = a.widths[blk];
        off = a.offsets[blk];
    }
    const SingleWhere<T> r = single_locate<T>(a, gi, W, off, a.widths != nullptr);
    if (r.err) single_error(a, r.err);
    T v0 = 0, v1 = 0;
    if (r.w0) { v0 = *r.w0; v1 = *r.w1; }
    out[k] = single_extract<T>(r, v0, v1);
}

// FOUR consecutive lookups per thread (round 6): one 32-byte read of indices, the four lookups' metadata / word loads issued together,
// one vector store.  With one lookup per thread a wavefront lived three dependent memory round trips for 64 results, and sorted or dense
// index vectors -- whose words hit in cache -- ran at 200-250 G lookups/s, a third of what their 8 + sizeof(T) bytes per lookup allow
// (profiles/r05_sweep_single.txt); scattered lookups get four fetches in flight per thread instead of one.
#ifndef FL_SINGLE_PER_THREAD
#define FL_SINGLE_PER_THREAD 4
#endif
constexpr unsigned SINGLE_PER_THREAD = FL_SINGLE_PER_THREAD;
template <typename T>
__global__ __launch_bounds__(WG) void k_unpack_single_x4(SingleArgs a)
{
    constexpr unsigned N = SINGLE_PER_THREAD;
    const uint64_t k = ((uint64_t)blockIdx.x * WG + threadIdx.x) * N;
    if (k + N > a.n_indices) return;                          // whole groups only (the launcher sends the tail to k_unpack_single)
    struct alignas(16) Idx2 { uint64_t v[2]; };
    const Idx2* ip = reinterpret_cast<const Idx2*>(a.indices + k);
    uint64_t gi[N];
    for (unsigned j = 0; j < N / 2; ++j) {
        const Idx2 two = ip[j];
        gi[2 * j] = two.v[0];
        gi[2 * j + 1] = two.v[1];
    }
    unsigned W[N];
    uint64_t off[N];
    bool in_range[N];
    for (unsigned j = 0; j < N; ++j) {
        W[j] = a.width;
        off[j] = 0;
        in_range[j] = (gi[j] >> 10) < a.n_blocks;
    }
    if (a.widths) {
        for (unsigned j = 0; j < N; ++j) {                    // all four blocks' metadata in flight together
            const uint64_t b = in_range[j] ? gi[j] >> 10 : 0;
            W[j] = a.widths[b];
            off[j] = a.offsets[b];
        }
    }
    SingleWhere<T> r[N];
    uint32_t err = 0;
    for (unsigned j = 0; j < N; ++j) {
        if (a.widths && !in_range[j]) { r[j] = SingleWhere<T>{nullptr, nullptr, 0u, 0u, 2u}; }   // bitpacking.rs:152 (before the width is known)
        else r[j] = single_locate<T>(a, gi[j], W[j], off[j], a.widths != nullptr);
        err |= r[j].err;
    }
    if (err) single_error(a, err);
    T v0[N], v1[N];
    for (unsigned j = 0; j < N; ++j) {                        // eight independent loads; a lookup without a word reads the column's first one
        const T* safe = static_cast<const T*>(a.packed);
        v0[j] = *(r[j].w0 ? r[j].w0 : safe);
        v1[j] = *(r[j].w0 ? r[j].w1 : safe);
    }
    struct alignas(sizeof(T) * N > 16 ? 16 : sizeof(T) * N) Out { T v[N]; };
    Out o;
    for (unsigned j = 0; j < N; ++j) o.v[j] = single_extract<T>(r[j], v0[j], v1[j]);
    *reinterpret_cast<Out*>(static_cast<T*>(a.out) + k) = o;
}

template <typename T>
hipError_t launch_unpack_single(const SingleArgs& a, hipStream_t s)
{
    if (a.n_indices == 0) return hipSuccess;
    // the vector form needs 16-byte aligned indices and a result pointer aligned to its 4-element store, and a column to read from
    const bool vec = (reinterpret_cast<uintptr_t>(a.indices) & 15u) == 0 && (reinterpret_cast<uintptr_t>(a.out) & ((sizeof(T) * SINGLE_PER_THREAD > 16 ? 16 : sizeof(T) * SINGLE_PER_THREAD) - 1)) == 0 &&
                     a.packed != nullptr && a.n_blocks > 0 && (a.widths ? a.packed_bytes >= sizeof(T) : a.width > 0);
    const uint64_t groups = vec ? a.n_indices / SINGLE_PER_THREAD : 0, done = groups * SINGLE_PER_THREAD;
    if (groups) {
        FL_LAUNCH((k_unpack_single_x4<T>), dim3((unsigned)((groups + WG - 1) / WG)), dim3(WG), 0, s, a);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    if (done < a.n_indices) {
        FL_LAUNCH((k_unpack_single<T>), dim3((unsigned)((a.n_indices - done + WG - 1) / WG)), dim3(WG), 0, s, a, done);
    }
    return hipGetLastError();
}

template <typename T> stream_launch_t transpose_launcher(bool inverse);
template <typename T> hipError_t unpack_single_launch(const SingleArgs& a, hipStream_t s);

}  // namespace fl
