// fl_scan.hpp -- widths[] -> offsets[] on the device (the exclusive prefix sum a caller of
// bitpacking.rs:109-129 keeps implicitly by advancing its packed slice by 128*W bytes per block).
// Included by exactly one translation unit of the library (fl_capi.hip), which instantiates what it uses.
#pragma once
#include "fl_kernels.hpp"

namespace fl {

// ---------------------------------------------------------------------------
// widths -> offsets: exclusive prefix sum of 128*widths[b] (bytes), entirely on the device and
// without scratch memory.  Three launches over chunks of SCAN_CHUNK blocks:
//   A: every chunk writes its LOCAL exclusive prefix; the slot of the chunk's first block (whose
//      local prefix is 0 by definition) temporarily holds the chunk TOTAL;
//   B: one workgroup turns the chunk totals (strided slots) into chunk base offsets, in place,
//      and writes the column's total packed bytes;
//   C: every chunk adds its base to its other slots.
// ---------------------------------------------------------------------------
constexpr int SCAN_PER_THREAD = 16;
constexpr int SCAN_CHUNK = WG * SCAN_PER_THREAD;   // 4096 blocks

struct ScanArgs {
    const uint8_t* widths;
    uint64_t* offsets;
    uint64_t* total;       // may be nullptr
    uint32_t* err_flag;    // may be nullptr
    uint64_t n_blocks;
    unsigned type_bits;
};

__device__ __forceinline__ uint64_t wave_incl_scan(uint64_t v, unsigned lane)
{
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t o = __shfl_up(v, d, 64);
        if (lane >= (unsigned)d) v += o;
    }
    return v;
}

// workgroup-wide exclusive scan of one value per thread; returns the exclusive prefix, *total = sum
__device__ __forceinline__ uint64_t wg_excl_scan(uint64_t v, uint64_t* total)
{
    __shared__ uint64_t wave_sum[WG / 64];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t inc = wave_incl_scan(v, lane);
    __syncthreads();                       // protect wave_sum against the previous call's readers
    if (lane == 63) wave_sum[wave] = inc;
    __syncthreads();
    uint64_t base = 0, tot = 0;
    for (unsigned k = 0; k < WG / 64; ++k) {
        if (k < wave) base += wave_sum[k];
        tot += wave_sum[k];
    }
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(WG) void k_scan_local(ScanArgs a)
{
    const uint64_t chunk0 = (uint64_t)blockIdx.x * SCAN_CHUNK;
    const uint64_t t0 = chunk0 + (uint64_t)threadIdx.x * SCAN_PER_THREAD;
    unsigned w[SCAN_PER_THREAD];
    uint64_t sum = 0;
    bool bad = false;
    for (int e = 0; e < SCAN_PER_THREAD; ++e) {
        w[e] = (t0 + e < a.n_blocks) ? a.widths[t0 + e] : 0u;
        bad |= w[e] > a.type_bits;
        sum += 128ull * w[e];
    }
    if (bad && a.err_flag) __hip_atomic_fetch_or(a.err_flag, 1u /* FL_DEVERR_WIDTH */, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint64_t total;
    uint64_t run = wg_excl_scan(sum, &total);
    for (int e = 0; e < SCAN_PER_THREAD; ++e) {
        if (t0 + e < a.n_blocks) a.offsets[t0 + e] = (threadIdx.x == 0 && e == 0) ? total : run;
        run += 128ull * w[e];
    }
}

__global__ __launch_bounds__(WG) void k_scan_chunks(ScanArgs a)
{
    const uint64_t n_chunks = (a.n_blocks + SCAN_CHUNK - 1) / SCAN_CHUNK;
    uint64_t carry = 0;
    for (uint64_t c0 = 0; c0 < n_chunks; c0 += WG) {
        const uint64_t c = c0 + threadIdx.x;
        const uint64_t v = c < n_chunks ? a.offsets[c * SCAN_CHUNK] : 0;
        uint64_t total;
        const uint64_t ex = wg_excl_scan(v, &total);
        if (c < n_chunks) a.offsets[c * SCAN_CHUNK] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0 && a.total) *a.total = carry;
}

__global__ __launch_bounds__(WG) void k_scan_add(ScanArgs a)
{
    const uint64_t chunk0 = (uint64_t)blockIdx.x * SCAN_CHUNK;
    const uint64_t base = a.offsets[chunk0];
    const uint64_t t0 = chunk0 + (uint64_t)threadIdx.x * SCAN_PER_THREAD;
    for (int e = 0; e < SCAN_PER_THREAD; ++e)
        if (t0 + e < a.n_blocks && !(threadIdx.x == 0 && e == 0)) a.offsets[t0 + e] += base;
}

inline hipError_t launch_widths_to_offsets(const ScanArgs& a, hipStream_t s)
{
    if (a.n_blocks == 0) {
        if (a.total) return hipMemsetAsync(a.total, 0, sizeof(uint64_t), s);
        return hipSuccess;
    }
    const unsigned n_chunks = (unsigned)((a.n_blocks + SCAN_CHUNK - 1) / SCAN_CHUNK);
    // three launches: FL_LAUNCH clears the error slot before each, so each one's status is read before the next
    FL_LAUNCH(k_scan_local, dim3(n_chunks), dim3(WG), 0, s, a);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    FL_LAUNCH(k_scan_chunks, dim3(1), dim3(WG), 0, s, a);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    FL_LAUNCH(k_scan_add, dim3(n_chunks), dim3(WG), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// An encoder's width choice for FoR with reference = the block's minimum (ffor.rs:24-36 packs `in[idx] - reference`, masked to
// W bits by macros.rs:73): widths[b] = number of bits of maxs[b] - mins[b] (0 when the block is constant) is the smallest W that
// loses nothing.  The reference has no width selection (closing note); this is the arithmetic its callers do
// between block_min_max and for_pack.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(WG) void k_for_widths(const T* mins, const T* maxs, uint64_t n_blocks, uint8_t* widths)
{
    const uint64_t b = (uint64_t)blockIdx.x * WG + threadIdx.x;
    if (b >= n_blocks) return;
    const T span = (T)(maxs[b] - mins[b]);                    // wrapping, like every subtraction on the path
    widths[b] = (uint8_t)(span == 0 ? 0 : 64 - __builtin_clzll((unsigned long long)span));
}

template <typename T>
hipError_t launch_for_widths(const T* mins, const T* maxs, uint64_t n_blocks, uint8_t* widths, hipStream_t s)
{
    if (n_blocks == 0) return hipSuccess;
    FL_LAUNCH((k_for_widths<T>), dim3((unsigned)((n_blocks + WG - 1) / WG)), dim3(WG), 0, s, mins, maxs, n_blocks, widths);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// fl_fill_random: counter-based test / benchmark data generated in HBM (SURVEY.md 8(d) "value distribution / seeds":
// splitmix64 of the global index, so a host can regenerate any part of the stream without a PCIe transfer).
// 64-bit word i = splitmix64 output number i+1 of the generator seeded with seed * GOLDEN -- the stream
// tests/datagen.py and the oracle's parallel fill produce.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void k_fill_splitmix64(uint64_t* dst, uint64_t n_words, uint64_t seed)
{
    const uint64_t GOLDEN = 0x9E3779B97F4A7C15ull;
    const uint64_t stride = (uint64_t)gridDim.x * WG;
    for (uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x; i < n_words; i += stride) {
        uint64_t z = seed * GOLDEN + (i + 1) * GOLDEN;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        __builtin_nontemporal_store(z ^ (z >> 31), dst + i);
    }
}

inline hipError_t launch_fill_random(uint64_t* dst, uint64_t n_words, uint64_t seed, hipStream_t s)
{
    if (n_words == 0) return hipSuccess;
    const uint64_t want = (n_words + WG - 1) / WG;
    FL_LAUNCH(k_fill_splitmix64, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(WG), 0, s, dst, n_words, seed);
    return hipGetLastError();
}

}  // namespace fl
