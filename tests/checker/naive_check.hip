// naive_check.hip -- TEST INFRASTRUCTURE (never linked into the product): a deliberately naive device-side checker of the
// codec's outputs, so that EVERY element of a column that fills the chip can be verified (tests/test_gpu_full_check.py).
//
// It shares no code with fastlanes_amd/csrc: one thread per VALUE (or per FL lane where the reference's body is stateful),
// plain scalar arithmetic written from the reference's closed forms --
//   * the wire format from `unpack_single` (bitpacking.rs:132-179 with the index tables of :207-232): value i of a block is
//     bits [row*W, row*W + W) of FL lane `lane`'s stream, lane = i % LANES, row = FL_ORDER[(i % 128 - lane) / 16] * 8 + i / 128,
//     stream word w of lane l at packed[LANES * w + l];
//   * index(row, lane) = FL_ORDER[row / 8] * 16 + (row % 8) * 128 + lane          (macros.rs:20-24)
//   * transpose(i) = (i % 16) * 64 + FL_ORDER[(i / 16) % 8] * 8 + i / 128         (transpose.rs:29-36)
//   * Delta per FL lane, rows in order, wrapping (delta.rs:24-63); FoR a wrapping add / subtract of one scalar (ffor.rs:24-50)
// -- and instead of producing a second copy of the output it COMPARES: every function adds the number of elements of `got`
// that differ from what the closed form says to *mismatches (device uint64).  tests/test_gpu_full_check.py validates the
// checker itself against the CPU oracle at small sizes (the oracle's output must count 0, a single flipped element 1).
//
// Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC tests/checker/naive_check.hip -o tests/checker/libfl_naive_check.so
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__device__ __constant__ const int ORDER[8] = {0, 4, 2, 6, 1, 5, 3, 7};   // lib.rs:22

enum Op { OP_UNPACK = 0, OP_PACK = 1, OP_DELTA = 2, OP_UNDELTA = 3, OP_UNDELTA_PACK = 4, OP_TRANSPOSE = 5, OP_UNTRANSPOSE = 6,
          OP_UNDELTA_PACK_UNTRANSPOSE = 7, OP_TRANSPOSE_DELTA_PACK = 8, OP_BLOCK_SUMS = 9, OP_COMPARE = 10, OP_MIN_MAX = 11 };

template <typename T> struct TT {
    static constexpr unsigned BITS = sizeof(T) * 8;
    static constexpr unsigned LANES = 1024 / BITS;
};

template <typename T> __device__ T low_mask(unsigned w) { return w >= TT<T>::BITS ? (T) ~(T)0 : (T)((((uint64_t)1) << w) - 1); }

__device__ unsigned index_of(unsigned row, unsigned lane) { return ORDER[row / 8] * 16 + (row % 8) * 128 + lane; }
__device__ unsigned transpose_of(unsigned i) { return (i % 16) * 64 + ORDER[(i / 16) % 8] * 8 + i / 128; }
template <typename T> __device__ void row_lane_of(unsigned i, unsigned& row, unsigned& lane)
{
    lane = i % TT<T>::LANES;
    const unsigned s = i / 128, o = (i % 128 - lane) / 16;
    row = ORDER[o] * 8 + s;
}

// bits [row*W, row*W + W) of FL lane `lane`'s stream (bitpacking.rs:164-178)
template <typename T> __device__ T field_of(const T* pk, unsigned w, unsigned row, unsigned lane)
{
    constexpr unsigned TB = TT<T>::BITS, L = TT<T>::LANES;
    if (w == 0) return 0;
    if (w == TB) return pk[L * row + lane];
    const unsigned bit = row * w, word = bit / TB, sh = bit % TB;
    uint64_t v = (uint64_t)pk[L * word + lane] >> sh;
    if (sh + w > TB) v |= (uint64_t)pk[L * (word + 1) + lane] << (TB - sh);
    return (T)v & low_mask<T>(w);
}
template <typename T> __device__ T single_of(const T* pk, unsigned w, unsigned i)
{
    unsigned row, lane;
    row_lane_of<T>(i, row, lane);
    return field_of<T>(pk, w, row, lane);
}

struct Args {
    int op;
    unsigned width;
    const void* a;        // first input (packed or unpacked, per op)
    const void* aux;      // FoR references [n * aux_stride] (may be null: no FoR) or Delta bases [n][LANES]
    uint64_t aux_stride;
    const void* got;      // the output under test
    const void* got2;     // block_min_max: maxs
    uint64_t n_blocks;
    unsigned long long* mismatches;
    int cmp_op;           // fl_cmp: 0 ==, 1 !=, 2 <, 3 <=, 4 >, 5 >=
    uint64_t cmp_k;
    const uint8_t* widths;      // mixed-width columns: per-block width / byte offset (else null)
    const uint64_t* offsets;
};

__device__ void count(unsigned long long* m, unsigned n) { if (n) atomicAdd(m, (unsigned long long)n); }

// one thread per VALUE: ops whose expected value is a closed form of the inputs
template <typename T> __global__ void k_per_value(Args g)
{
    constexpr unsigned TB = TT<T>::BITS, L = TT<T>::LANES;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= g.n_blocks * 1024) return;
    const uint64_t b = t / 1024;
    const unsigned i = (unsigned)(t % 1024);
    unsigned w = g.width;
    uint64_t pk_off = b * (uint64_t)(1024 * w / TB);
    if (g.widths) { w = g.widths[b]; pk_off = g.offsets[b] / sizeof(T); }
    const T* a = static_cast<const T*>(g.a);
    const T* got = static_cast<const T*>(g.got);
    const T* aux = static_cast<const T*>(g.aux);
    unsigned row, lane;
    row_lane_of<T>(i, row, lane);
    bool bad = false;
    switch (g.op) {
    case OP_UNPACK: {                                   // unpack / unfor_pack (ffor.rs:46-48)
        T want = single_of<T>(a + pk_off, w, i);
        if (aux) want = (T)(want + aux[b * g.aux_stride]);
        bad = got[b * 1024 + i] != want;
        break;
    }
    case OP_PACK: {                                     // pack / for_pack: every field of the packed block (= every bit of it)
        if (w == 0) break;                              // macros.rs:52-53: nothing is written
        T v = a[b * 1024 + i];
        if (aux) v = (T)(v - aux[b * g.aux_stride]);    // ffor.rs:32-34
        if (w < TB) v &= low_mask<T>(w);                // macros.rs:73; W == T copies unmasked (:54-59)
        bad = single_of<T>(got + pk_off, w, i) != v;
        break;
    }
    case OP_DELTA: {                                    // delta.rs:28-30
        const T prev = row == 0 ? aux[b * L + lane] : a[b * 1024 + index_of(row - 1, lane)];
        bad = got[b * 1024 + i] != (T)(a[b * 1024 + i] - prev);
        break;
    }
    case OP_TRANSPOSE: bad = got[b * 1024 + i] != a[b * 1024 + transpose_of(i)]; break;                 // transpose.rs:12-14
    case OP_UNTRANSPOSE: bad = got[b * 1024 + transpose_of(i)] != a[b * 1024 + i]; break;               // transpose.rs:19-21
    case OP_TRANSPOSE_DELTA_PACK: {                     // pack::<W>(delta(transpose(v), base))  (delta.rs:88-95)
        if (w == 0) break;
        const T cur = a[b * 1024 + transpose_of(i)];
        const T prev = row == 0 ? aux[b * L + lane] : a[b * 1024 + transpose_of(index_of(row - 1, lane))];
        T d = (T)(cur - prev);
        if (w < TB) d &= low_mask<T>(w);
        bad = single_of<T>(got + pk_off, w, i) != d;
        break;
    }
    case OP_COMPARE: {                                  // mask bit i of block b = (unpack(block)[i] <op> k)
        const T x = single_of<T>(a + pk_off, w, i), k = (T)g.cmp_k;
        const bool want = g.cmp_op == 0 ? x == k : g.cmp_op == 1 ? x != k : g.cmp_op == 2 ? x < k : g.cmp_op == 3 ? x <= k
                        : g.cmp_op == 4 ? x > k : x >= k;
        const uint32_t word = static_cast<const uint32_t*>(g.got)[b * 32 + i / 32];
        bad = (((word >> (i % 32)) & 1u) != 0) != want;
        break;
    }
    default: break;
    }
    if (bad) count(g.mismatches, 1);
}

// one thread per FL LANE: the stateful bodies -- rows in order, a running value (delta.rs:36-45, :47-63)
template <typename T> __global__ void k_per_lane(Args g)
{
    constexpr unsigned TB = TT<T>::BITS, L = TT<T>::LANES;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= g.n_blocks * L) return;
    const uint64_t b = t / L;
    const unsigned lane = (unsigned)(t % L);
    unsigned w = g.width;
    uint64_t pk_off = b * (uint64_t)(1024 * w / TB);
    if (g.widths) { w = g.widths[b]; pk_off = g.offsets[b] / sizeof(T); }
    const T* a = static_cast<const T*>(g.a);
    const T* got = static_cast<const T*>(g.got);
    const T* pk = a + pk_off;
    T run = static_cast<const T*>(g.aux)[b * L + lane];
    unsigned bad = 0;
    for (unsigned row = 0; row < TB; ++row) {
        const unsigned idx = index_of(row, lane);
        const T elem = g.op == OP_UNDELTA ? a[b * 1024 + idx] : field_of<T>(pk, w, row, lane);
        run = (T)(run + elem);
        const unsigned at = g.op == OP_UNDELTA_PACK_UNTRANSPOSE ? transpose_of(idx) : idx;   // untranspose: out[transpose(i)] = in[i]
        bad += got[b * 1024 + at] != run;
    }
    count(g.mismatches, bad);
}

// one thread per BLOCK: reductions over the 1024 values
template <typename T> __global__ void k_per_block(Args g)
{
    constexpr unsigned TB = TT<T>::BITS;
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= g.n_blocks) return;
    const unsigned w = g.width;
    if (g.op == OP_BLOCK_SUMS) {
        const T* pk = static_cast<const T*>(g.a) + b * (uint64_t)(1024 * w / TB);
        uint64_t s = 0;
        for (unsigned i = 0; i < 1024; ++i) s += single_of<T>(pk, w, i);
        count(g.mismatches, static_cast<const uint64_t*>(g.got)[b] != s);
    } else {
        const T* v = static_cast<const T*>(g.a) + b * 1024;
        T mn = v[0], mx = v[0];
        for (unsigned i = 1; i < 1024; ++i) { mn = v[i] < mn ? v[i] : mn; mx = v[i] > mx ? v[i] : mx; }
        count(g.mismatches, (static_cast<const T*>(g.got)[b] != mn) + (static_cast<const T*>(g.got2)[b] != mx));
    }
}

template <typename T> int run(const Args& g, hipStream_t s)
{
    if (g.n_blocks == 0) return 0;
    const bool per_lane = g.op == OP_UNDELTA || g.op == OP_UNDELTA_PACK || g.op == OP_UNDELTA_PACK_UNTRANSPOSE;
    const bool per_block = g.op == OP_BLOCK_SUMS || g.op == OP_MIN_MAX;
    const uint64_t threads = per_block ? g.n_blocks : per_lane ? g.n_blocks * TT<T>::LANES : g.n_blocks * 1024;
    const uint64_t grid = (threads + 255) / 256;
    if (grid > 0x7fffffffull) return (int)hipErrorInvalidValue;
    if (per_block) hipLaunchKernelGGL(k_per_block<T>, dim3((unsigned)grid), dim3(256), 0, s, g);
    else if (per_lane) hipLaunchKernelGGL(k_per_lane<T>, dim3((unsigned)grid), dim3(256), 0, s, g);
    else hipLaunchKernelGGL(k_per_value<T>, dim3((unsigned)grid), dim3(256), 0, s, g);
    return (int)hipGetLastError();
}

}  // namespace

// int naive_check(type_bits, op, width, a, aux, aux_stride, got, got2, n_blocks, mismatches, cmp_op, cmp_k, widths, offsets, stream)
extern "C" int naive_check(unsigned type_bits, int op, unsigned width, const void* a, const void* aux, uint64_t aux_stride, const void* got,
                           const void* got2, uint64_t n_blocks, unsigned long long* mismatches, int cmp_op, uint64_t cmp_k,
                           const uint8_t* widths, const uint64_t* offsets, void* stream)
{
    if (width > type_bits || op < 0 || op > OP_MIN_MAX) return -1;
    const Args g{op, width, a, aux, aux_stride, got, got2, n_blocks, mismatches, cmp_op, cmp_k, widths, offsets};
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (type_bits) {
    case 8: return run<uint8_t>(g, s);
    case 16: return run<uint16_t>(g, s);
    case 32: return run<uint32_t>(g, s);
    case 64: return run<uint64_t>(g, s);
    default: return -1;
    }
}
