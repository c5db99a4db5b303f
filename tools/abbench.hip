// abbench.hip -- within-process interleaved A/B of launch-shape / cache-policy variants
// of the u32 W=7 unpack kernel, next to plain read / write / copy streams of the same
// byte mix (the in-situ ceilings).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17
// -I fastlanes_amd/csrc tools/abbench.hip -o tools/abbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#include "fl_device.hpp"

using namespace fl;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Args { const u32x4* in; u32x4* out; uint64_t n_blocks; };

// MODE: 0 one-shot, 1 grid-stride, 2 XCD-contiguous remap (blockIdx%8 selects one of 8 contiguous chunks)
template <typename T, int W, int WGS, bool NTL, bool NTS, int MODE>
__global__ __launch_bounds__(WGS) void k_unpack_v(Args a)
{
    constexpr int BPW = WGS / 8;
    const uint64_t n_wg = (a.n_blocks + BPW - 1) / BPW;
    uint64_t wg = blockIdx.x;
    if (MODE == 2) {
        const uint64_t per = (n_wg + 7) / 8;
        wg = (uint64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if (wg >= n_wg || (blockIdx.x >> 3) >= per) return;
    }
    for (; wg < n_wg; wg += (MODE == 1 ? gridDim.x : n_wg)) {
        const uint64_t blk = wg * BPW + (threadIdx.x >> 3);
        const unsigned c = threadIdx.x & 7u;
        if (blk < a.n_blocks) {
            Cell<T> in[W];
            const u32x4* pk = a.in + blk * (uint64_t)(8 * W) + c;
            static_for<W>([&](auto Wd) { in[decltype(Wd)::value] = load_cell<T, NTL>(pk + 8 * decltype(Wd)::value); });
            u32x4* un = a.out + blk * (uint64_t)Elem<T>::CELLS_PER_BLOCK + c;
            unpack_rows<T, W>(in, [&](auto R, const Cell<T>& v) {
                store_cell<T, NTS>(un + Elem<T>::row_cell(decltype(R)::value), v);
            });
        }
    }
}

// CHUNK-granular XCD remap + buffer-store cache policy (AUX: 0 plain, 1 sc0, 2 nt, 16 sc1, combos; -1 = global store)
// blockIdx b runs on XCD b%8 (observed).  Remap so that each XCD owns runs of K consecutive workgroups:
//   wg = (b / (8K)) * 8K + (b % 8) * K + (b / 8) % K
template <typename T, int W, int AUX, int LAUX, int MAXW = 8, int ORDER = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, MAXW))) void k_unpack_x(Args a, unsigned K)
{
    constexpr int BPW = 32;
    const uint64_t n_wg = (a.n_blocks + BPW - 1) / BPW;
    const uint64_t b = blockIdx.x;
    uint64_t wg = b;
    if (K > 1) {
        const uint64_t span = 8ull * K;                   // grid is rounded up to whole spans
        wg = (b / span) * span + (b % 8) * K + (b / 8) % K;
    }
    if (wg >= n_wg) return;
    const unsigned tid = threadIdx.x;
    const uint64_t blk = wg * BPW + (tid >> 3);
    const unsigned c = tid & 7u;
    if (blk >= a.n_blocks) return;
    Cell<T> in[W];
    if constexpr (LAUX < 0) {
        const u32x4* pk = a.in + blk * (uint64_t)(8 * W) + c;
        static_for<W>([&](auto Wd) { in[decltype(Wd)::value] = load_cell<T, false>(pk + 8 * decltype(Wd)::value); });
    } else {
        const u32x4* wg_in = a.in + wg * (uint64_t)(BPW * 8 * W);
        auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)wg_in, 0, BPW * 128 * W, 0x00020000);
        const unsigned vo = (tid >> 3) * (128 * W) + c * 16;
        static_for<W>([&](auto Wd) {
            in[decltype(Wd)::value] = __builtin_bit_cast(Cell<T>, __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 128 * decltype(Wd)::value, 0, LAUX));
        });
    }
    if constexpr (AUX < 0) {
        u32x4* un = a.out + blk * (uint64_t)Elem<T>::CELLS_PER_BLOCK + c;
        unpack_rows<T, W>(in, [&](auto R, const Cell<T>& v) { store_cell<T, false>(un + Elem<T>::row_cell(decltype(R)::value), v); });
    } else {
        u32x4* wg_out = a.out + wg * (uint64_t)(BPW * Elem<T>::CELLS_PER_BLOCK);
        auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)wg_out, 0, BPW * Elem<T>::CELLS_PER_BLOCK * 16, 0x00020000);
        const unsigned vo = (tid >> 3) * (Elem<T>::CELLS_PER_BLOCK * 16) + c * 16;
        if constexpr (ORDER == 0) {
            unpack_rows<T, W>(in, [&](auto R, const Cell<T>& v) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, vo + 16 * Elem<T>::row_cell(decltype(R)::value), 0, AUX);
            });
        } else {
            // ascending address order: cell-row j*sizeof(T) holds logical row (j/8 -> s, j%8 -> FL_ORDER^-1)
            static_for<Elem<T>::BITS>([&](auto J) {
                constexpr int j = decltype(J)::value;
                constexpr int per_s = Elem<T>::BITS / 8;                 // rows per s-group
                constexpr int s_ = j / per_s, f_ = j % per_s;            // f_ = FL_ORDER[o] rank
                // find o with fl_order(o) == f_-th smallest among o < per_s
                constexpr int o = fl_order(f_ * (8 / per_s)) ;          // FL_ORDER is self-inverse
                constexpr int r = o * 8 + s_;
                const Cell<T> v = unpack_row<T, W, r>(in);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, vo + 16 * Elem<T>::row_cell(r), 0, AUX);
            });
        }
    }
}

template <typename T, int W, int AUX, int LAUX, int MAXW = 8>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, MAXW))) void k_pack_x(Args a, unsigned K)
{
    constexpr int BPW = 32;
    constexpr int TB = Elem<T>::BITS;
    const uint64_t b = blockIdx.x;
    uint64_t wg = b;
    if (K > 1) { const uint64_t span = 8ull * K; wg = (b / span) * span + (b % 8) * K + (b / 8) % K; }
    const uint64_t n_wg = (a.n_blocks + BPW - 1) / BPW;
    if (wg >= n_wg) return;
    const unsigned tid = threadIdx.x;
    const uint64_t blk = wg * BPW + (tid >> 3);
    const unsigned c = tid & 7u;
    if (blk >= a.n_blocks) return;
    Cell<T> rows[TB];
    if constexpr (LAUX < 0) {
        const u32x4* un = a.in + blk * (uint64_t)Elem<T>::CELLS_PER_BLOCK + c;
        static_for<TB>([&](auto R) { rows[decltype(R)::value] = load_cell<T, false>(un + Elem<T>::row_cell(decltype(R)::value)); });
    } else {
        const u32x4* wg_in = a.in + wg * (uint64_t)(BPW * Elem<T>::CELLS_PER_BLOCK);
        auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)wg_in, 0, BPW * Elem<T>::CELLS_PER_BLOCK * 16, 0x00020000);
        const unsigned vo = (tid >> 3) * (Elem<T>::CELLS_PER_BLOCK * 16) + c * 16;
        static_for<TB>([&](auto R) {
            rows[decltype(R)::value] = __builtin_bit_cast(Cell<T>, __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 16 * Elem<T>::row_cell(decltype(R)::value), 0, LAUX));
        });
    }
    u32x4* wg_out = a.out + wg * (uint64_t)(BPW * 8 * W);
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)wg_out, 0, BPW * 128 * W, 0x00020000);
    const unsigned vo = (tid >> 3) * (128 * W) + c * 16;
    u32x4* pk = a.out + blk * (uint64_t)(8 * W) + c;
    pack_rows<T, W>([&](auto R) { return rows[decltype(R)::value]; },
                    [&](auto Wd, const Cell<T>& v) {
                        if constexpr (AUX < 0) store_cell<T, false>(pk + 8 * decltype(Wd)::value, v);
                        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, vo + 128 * decltype(Wd)::value, 0, AUX);
                    });
}

// Persistent XCD-aware stream: grid = 8 * wgs_per_xcd resident workgroups; XCD x (= blockIdx%8) owns the
// contiguous eighth [x*per, (x+1)*per) of the workgroup-tiles and its workgroups stride through it together,
// so each XCD's in-flight window is wgs_per_xcd consecutive 128-KiB tiles.  Next tile's packed words are
// prefetched before the current tile's stores are issued (vmcnt is in-order for loads and stores).
template <typename T, int W, int AUX, bool PREFETCH>
__global__ __launch_bounds__(256) void k_unpack_p(Args a, unsigned wgs_per_xcd)
{
    constexpr int BPW = 32;
    const uint64_t n_wg = (a.n_blocks + BPW - 1) / BPW;
    const uint64_t per = (n_wg + 7) / 8;
    const unsigned x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const uint64_t lo = x * per, hi = (lo + per < n_wg) ? lo + per : n_wg;
    const unsigned tid = threadIdx.x;
    const unsigned c = tid & 7u;
    uint64_t wg = lo + j;
    if (wg >= hi) return;
    Cell<T> cur[W], nxt[W];
    auto load_tile = [&](uint64_t w_, Cell<T>* dst) {
        const uint64_t blk = w_ * BPW + (tid >> 3);
        if (blk < a.n_blocks) {
            const u32x4* pk = a.in + blk * (uint64_t)(8 * W) + c;
            static_for<W>([&](auto Wd) { dst[decltype(Wd)::value] = load_cell<T, false>(pk + 8 * decltype(Wd)::value); });
        }
    };
    load_tile(wg, cur);
    for (; wg < hi; wg += wgs_per_xcd) {
        const uint64_t nw = wg + wgs_per_xcd;
        if (PREFETCH && nw < hi) load_tile(nw, nxt);
        const uint64_t blk = wg * BPW + (tid >> 3);
        if (blk < a.n_blocks) {
            u32x4* wg_out = a.out + wg * (uint64_t)(BPW * Elem<T>::CELLS_PER_BLOCK);
            auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)wg_out, 0, BPW * Elem<T>::CELLS_PER_BLOCK * 16, 0x00020000);
            const unsigned vo = (tid >> 3) * (Elem<T>::CELLS_PER_BLOCK * 16) + c * 16;
            unpack_rows<T, W>(cur, [&](auto R, const Cell<T>& v) {
                if constexpr (AUX < 0) store_cell<T, false>(wg_out + (tid >> 3) * Elem<T>::CELLS_PER_BLOCK + c + Elem<T>::row_cell(decltype(R)::value), v);
                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, vo + 16 * Elem<T>::row_cell(decltype(R)::value), 0, AUX);
            });
        }
        if (nw < hi) {
            if (PREFETCH) { static_for<W>([&](auto Wd) { cur[decltype(Wd)::value] = nxt[decltype(Wd)::value]; }); }
            else load_tile(nw, cur);
        }
    }
}

// cell-column compute + LDS-staged WAVE-CONTIGUOUS stores: every group of 8 address-rows
// (1 KiB per block) goes through LDS so that each store instruction writes 1 KiB contiguous
// bytes of ONE block (as the wave-per-block sketch does) instead of 8 x 128 B at 4 KiB stride.
template <typename T, int W, int AUX, int MAXW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, MAXW))) void k_unpack_cc_wavestore(Args a, unsigned K)
{
    constexpr int BPW = 32;
    constexpr int TB = Elem<T>::BITS;
    constexpr int GROUPS = TB / 8;                         // 1 KiB per block per group
    __shared__ __attribute__((aligned(16))) char lds[4][8 * 1024];
    const uint64_t n_wg = (a.n_blocks + BPW - 1) / BPW;
    const uint64_t b = blockIdx.x;
    uint64_t wg = b;
    if (K > 1) { const uint64_t span = 8ull * K; wg = (b / span) * span + (b % 8) * K + (b / 8) % K; }
    if (wg >= n_wg) return;
    const unsigned tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const unsigned g = lane >> 3, c = lane & 7u;
    const uint64_t blk = wg * BPW + wave * 8 + g;
    Cell<T> in[W];
    if (blk < a.n_blocks) {
        const u32x4* pk = a.in + blk * (uint64_t)(8 * W) + c;
        static_for<W>([&](auto Wd) { in[decltype(Wd)::value] = load_cell<T, false>(pk + 8 * decltype(Wd)::value); });
    }
    // the wave's 8 blocks are one contiguous 8*BLOCK_BYTES region
    constexpr unsigned BLOCK_BYTES = Elem<T>::CELLS_PER_BLOCK * 16;
    const uint64_t first_blk = wg * BPW + wave * 8;
    const uint64_t rem = a.n_blocks > first_blk ? a.n_blocks - first_blk : 0;
    auto rs = __builtin_amdgcn_make_buffer_rsrc((char*)a.out + first_blk * BLOCK_BYTES, 0,
                                                (unsigned)(rem < 8 ? rem : 8) * BLOCK_BYTES, 0x00020000);
    char* my = &lds[wave][0];
    constexpr int PER_S = TB / 8;
    static_for<GROUPS>([&](auto KK) {
        constexpr int k = decltype(KK)::value;
        static_for<8>([&](auto I) {
            constexpr int j = 8 * k + decltype(I)::value;                     // address-row
            constexpr int row = fl_order((j % PER_S) * (8 / PER_S)) * 8 + j / PER_S;
            const Cell<T> v = unpack_row<T, W, row>(in);
            *reinterpret_cast<u32x4*>(my + g * 1024 + decltype(I)::value * 128 + c * 16) = __builtin_bit_cast(u32x4, v);
        });
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        static_for<8>([&](auto B) {
            constexpr int bb = decltype(B)::value;
            const u32x4 v = *reinterpret_cast<const u32x4*>(my + bb * 1024 + lane * 16);
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, bb * BLOCK_BYTES + k * 1024 + lane * 16, 0, AUX);
        });
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    });
}

// The north_star sketch, for the record: a wavefront owns whole blocks (here 2: lanes 0-31 /
// 32-63 = the 32 FL lanes of block A / B), each lane unpacks its FL lane's 32 values with
// compile-time shifts from 4-byte loads, the values are scattered to their index(row,lane)
// position in LDS (ds_write_b32, conflict-free) and read back linearly (ds_read_b128) so the
// global stores are fully contiguous 1 KiB-per-instruction dwordx4.  Same XCD map, st18, maxw.
template <int W, int AUX, int MAXW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, MAXW))) void k_unpack_waveblock(Args a, unsigned K)
{
    using T = uint32_t;
    constexpr int BPW = 8;                       // 4 waves x 2 blocks
    __shared__ __attribute__((aligned(16))) uint32_t lds[4][2 * 1024];
    const uint64_t b = blockIdx.x;
    uint64_t wg = b;
    if (K > 1) { const uint64_t span = 8ull * K; wg = (b / span) * span + (b % 8) * K + (b / 8) % K; }
    const uint64_t n_wg = (a.n_blocks + BPW - 1) / BPW;
    if (wg >= n_wg) return;
    const unsigned tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint64_t blk = wg * BPW + wave * 2 + (lane >> 5);
    const unsigned l = lane & 31u;
    if (blk < a.n_blocks) {
        const uint32_t* pk = reinterpret_cast<const uint32_t*>(a.in) + blk * (uint64_t)(32 * W) + l;
        uint32_t w[W];
        static_for<W>([&](auto I) { w[decltype(I)::value] = pk[32 * decltype(I)::value]; });
        uint32_t* dst = &lds[wave][(lane >> 5) * 1024];
        static_for<32>([&](auto R) {
            constexpr int r = decltype(R)::value;
            constexpr int curr = r * W / 32, next = (r + 1) * W / 32, sh = r * W % 32;
            uint32_t v;
            if constexpr (next > curr && next < W && ((r + 1) * W % 32) > 0)
                v = ((w[curr] >> sh) | (w[next] << (32 - sh))) & ((1u << W) - 1u);
            else
                v = (w[curr] >> sh) & ((1u << W) - 1u);
            dst[fl_order(r / 8) * 16 + (r % 8) * 128 + l] = v;
        });
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // 2 blocks x 4 KiB per wave = 8 x 1 KiB store instructions
    u32x4* wave_out = a.out + (wg * BPW + wave * 2) * 256ull;
    const uint64_t rem_blocks = a.n_blocks > wg * BPW + wave * 2 ? a.n_blocks - (wg * BPW + wave * 2) : 0;
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)wave_out, 0, (unsigned)(rem_blocks < 2 ? rem_blocks : 2) * 4096, 0x00020000);
    static_for<8>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const u32x4 v = *reinterpret_cast<const u32x4*>(&lds[wave][(i * 64 + lane) * 4]);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (i * 64 + lane) * 16, 0, AUX);
    });
}

// plain streams with the same per-thread shape: RD cells read, WR cells written per thread
template <int RD, int WR, bool NT>
__global__ __launch_bounds__(256) void k_stream(const u32x4* in, u32x4* out, uint64_t n_threads, u32x4* sink)
{
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n_threads) return;
    const uint64_t wave = g >> 6; const unsigned lane = g & 63;
    u32x4 acc = {0, 0, 0, 0};
    // wave-contiguous 1 KiB accesses
#pragma unroll
    for (int i = 0; i < RD; ++i) {
        const u32x4* p = in + (wave * RD + i) * 64 + lane;
        acc += NT ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        u32x4 v = acc + (unsigned)i;
        u32x4* p = out + (wave * WR + i) * 64 + lane;
        if (NT) __builtin_nontemporal_store(v, p); else *p = v;
    }
    if (WR == 0 && acc.x == 0x12345678u) *sink = acc;
}

__global__ void k_fill(uint64_t* p, uint64_t n);

// tuned plain streams: XCD-contiguous tiles, sc1|nt buffer stores, waves/SIMD cap -- the ceiling of
// the memory system for a given read:write mix under the same policy as the product kernels
template <int RD, int WR, int MAXW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, MAXW)))
void k_stream_tuned(const u32x4* in, u32x4* out, uint64_t n_tiles, uint64_t tiles_per_xcd, u32x4* sink)
{
    const uint64_t tile = (uint64_t)(blockIdx.x & 7u) * tiles_per_xcd + (blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    const unsigned tid = threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    const u32x4* src = in + tile * (uint64_t)(256 * RD);
#pragma unroll
    for (int i = 0; i < RD; ++i) acc += src[i * 256 + tid];
    if constexpr (WR > 0) {
        auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(out + tile * (uint64_t)(256 * WR)), 0, 256 * WR * 16, 0x00020000);
#pragma unroll
        for (int i = 0; i < WR; ++i) __builtin_amdgcn_raw_buffer_store_b128(acc + (unsigned)i, rs, (i * 256 + tid) * 16, 0, 18);
    } else if (acc.x == 0x12345678u) *sink = acc;
}

struct Variant { std::string name; double bytes; std::function<void()> launch; std::vector<float> ms; };

int main(int argc, char** argv)
{
    const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 10000000ull;
    const int rounds = argc > 2 ? atoi(argv[2]) : 7;
    u32x4 *in, *out, *sink;
    CK(hipMalloc(&in, n * 1088 + (128 << 20)));
    CK(hipMalloc(&out, n * 4096 + (512 << 20)));
    CK(hipMalloc(&sink, 64));
    // random fill on device (never zero data: DVFS)
    hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, 0, (uint64_t*)in, (n * 1088 + (128 << 20)) / 8);
    hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, 0, (uint64_t*)out, n * 4096 / 8);
    CK(hipDeviceSynchronize());

    Args a{in, out, n};
    std::vector<Variant> vs;
    const double bytes = (double)n * 4992;
    const uint64_t n_wg0 = (n + 31) / 32;
    const unsigned G = (unsigned)((n_wg0 + 7) / 8);   // giant chunk: one contiguous eighth per XCD
    auto addX = [&](const std::string& name, auto kern, unsigned K, Args aa, double by, unsigned lds) {
        uint64_t n_wg = n_wg0;
        if (K > 1) n_wg = (n_wg + 8ull * K - 1) / (8ull * K) * (8ull * K);
        vs.push_back({name, by, [=]() { hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(256), lds, 0, aa, K); }, {}});
    };
    addX("cell-column base (plain)", k_unpack_x<uint32_t, 7, -1, -1>, 1, a, bytes, 0);
    addX("cell-column G st18 maxw2 addr-order", k_unpack_x<uint32_t, 7, 18, -1, 2, 1>, G, a, bytes, 0);
    {
        const uint64_t n_wg8 = (n + 7) / 8;
        const unsigned G8 = (unsigned)((n_wg8 + 7) / 8);
        auto addW = [&](const std::string& name, auto kern, unsigned K) {
            uint64_t n_wg = n_wg8;
            if (K > 1) n_wg = (n_wg + 8ull * K - 1) / (8ull * K) * (8ull * K);
            vs.push_back({name, bytes, [=]() { hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(256), 0, 0, a, K); }, {}});
        };
        addW("wave-per-block+LDS plain K=1 maxw8", k_unpack_waveblock<7, 0, 8>, 1);
        addW("wave-per-block+LDS G st18 maxw2", k_unpack_waveblock<7, 18, 2>, G8);
        addW("wave-per-block+LDS G st18 maxw4", k_unpack_waveblock<7, 18, 4>, G8);
        addW("wave-per-block+LDS G st18 maxw8", k_unpack_waveblock<7, 18, 8>, G8);
    }
    addX("cell-column + LDS wave-contig stores maxw2", k_unpack_cc_wavestore<uint32_t, 7, 18, 2>, G, a, bytes, 0);
    addX("cell-column + LDS wave-contig stores maxw3", k_unpack_cc_wavestore<uint32_t, 7, 18, 3>, G, a, bytes, 0);
    addX("cell-column + LDS wave-contig stores maxw4", k_unpack_cc_wavestore<uint32_t, 7, 18, 4>, G, a, bytes, 0);
    addX("cell-column + LDS wave-contig stores maxw5", k_unpack_cc_wavestore<uint32_t, 7, 18, 5>, G, a, bytes, 0);
    addX("cell-column + LDS wave-contig stores maxw8", k_unpack_cc_wavestore<uint32_t, 7, 18, 8>, G, a, bytes, 0);
    addX("cell-column + LDS wave-contig stores maxw3 (again)", k_unpack_cc_wavestore<uint32_t, 7, 18, 3>, G, a, bytes, 0);
    addX("cell-column G st18 maxw2 addr-order (again)", k_unpack_x<uint32_t, 7, 18, -1, 2, 1>, G, a, bytes, 0);
    const uint64_t n_thr = n * 8;
    auto addS = [&](const char* name, auto kern, double by) {
        vs.push_back({name, by, [=]() { hipLaunchKernelGGL(kern, dim3((unsigned)((n_thr + 255) / 256)), dim3(256), 0, 0, (const u32x4*)in, out, n_thr, sink); }, {}});
    };
    {
        const uint64_t n_tiles = n / 32;   // a tile = 256 threads x (RD+WR) cells = 32 blocks' worth
        const uint64_t tpx = (n_tiles + 7) / 8;
        auto addT = [&](const char* name, auto kern, double by) {
            vs.push_back({name, by, [=]() { hipLaunchKernelGGL(kern, dim3((unsigned)(tpx * 8)), dim3(256), 0, 0, (const u32x4*)in, out, n_tiles, tpx, sink); }, {}});
        };
        addT("tuned stream 7rd:32wr maxw2", k_stream_tuned<7, 32, 2>, (double)n_tiles * 32 * 4992);
        addT("tuned stream 7rd:32wr maxw4", k_stream_tuned<7, 32, 4>, (double)n_tiles * 32 * 4992);
        addT("tuned stream write-only 32 maxw2", k_stream_tuned<0, 32, 2>, (double)n_tiles * 32 * 4096);
        addT("tuned stream write-only 32 maxw8", k_stream_tuned<0, 32, 8>, (double)n_tiles * 32 * 4096);
        addT("tuned stream read-only 7 maxw2", k_stream_tuned<7, 0, 2>, (double)n_tiles * 32 * 896);
        addT("tuned stream read-only 7 maxw8", k_stream_tuned<7, 0, 8>, (double)n_tiles * 32 * 896);
        // read-heavy mixes read the big buffer (`out`, 4096 B/block) and write the small one (`in`)
        auto addR = [&](const char* name, auto kern, double by) {
            vs.push_back({name, by, [=]() { hipLaunchKernelGGL(kern, dim3((unsigned)(tpx * 8)), dim3(256), 0, 0, (const u32x4*)out, in, n_tiles, tpx, sink); }, {}});
        };
        addR("tuned stream read-only 32 maxw1", k_stream_tuned<32, 0, 1>, (double)n_tiles * 32 * 4096);
        addR("tuned stream read-only 32 maxw2", k_stream_tuned<32, 0, 2>, (double)n_tiles * 32 * 4096);
        addR("tuned stream 32rd:7wr maxw1", k_stream_tuned<32, 7, 1>, (double)n_tiles * 32 * 4992);
        addR("tuned stream 32rd:7wr maxw2", k_stream_tuned<32, 7, 2>, (double)n_tiles * 32 * 4992);
    }
    addS("stream 7rd:32wr nt", k_stream<7, 32, true>, bytes);
    addS("stream write-only 32", k_stream<0, 32, false>, (double)n * 4096);

    // the two designs must produce identical bytes (checked on the first and last 16 blocks)
    {
        std::vector<uint32_t> ref(2 * 16 * 1024), alt(2 * 16 * 1024);
        const uint64_t G_ = G;
        auto grab = [&](std::vector<uint32_t>& h) {
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), out, 16 * 4096, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h.data() + 16 * 1024, (char*)out + (n - 16) * 4096, 16 * 4096, hipMemcpyDeviceToHost));
        };
        CK(hipMemset(out, 0xAB, n * 4096));
        uint64_t n_wg = (n_wg0 + 8ull * G_ - 1) / (8ull * G_) * (8ull * G_);
        hipLaunchKernelGGL((k_unpack_x<uint32_t, 7, 18, -1, 2, 1>), dim3((unsigned)n_wg), dim3(256), 0, 0, a, (unsigned)G_);
        grab(ref);
        CK(hipMemset(out, 0xCD, n * 4096));
        const uint64_t n_wg8 = (n + 7) / 8;
        const unsigned G8 = (unsigned)((n_wg8 + 7) / 8);
        n_wg = (n_wg8 + 8ull * G8 - 1) / (8ull * G8) * (8ull * G8);
        hipLaunchKernelGGL((k_unpack_waveblock<7, 18, 2>), dim3((unsigned)n_wg), dim3(256), 0, 0, a, G8);
        grab(alt);
        printf("wave-per-block+LDS output == cell-column output on sampled blocks: %s\n", ref == alt ? "yes" : "NO");
        CK(hipMemset(out, 0xEF, n * 4096));
        n_wg = (n_wg0 + 8ull * G_ - 1) / (8ull * G_) * (8ull * G_);
        hipLaunchKernelGGL((k_unpack_cc_wavestore<uint32_t, 7, 18, 2>), dim3((unsigned)n_wg), dim3(256), 0, 0, a, (unsigned)G_);
        grab(alt);
        printf("cell-column + LDS wave-contig stores output == cell-column output: %s\n", ref == alt ? "yes" : "NO");
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& v : vs) { v.launch(); }
    CK(hipDeviceSynchronize());
    for (int r = 0; r < rounds; ++r)
        for (auto& v : vs) {
            CK(hipEventRecord(e0, 0));
            v.launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            v.ms.push_back(ms);
        }
    CK(hipGetLastError());
    printf("%-36s %9s %9s %9s %9s\n", "variant", "med_ms", "min_ms", "GB/s_med", "GB/s_max");
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        float med = v.ms[v.ms.size() / 2], mn = v.ms[0];
        printf("%-36s %9.4f %9.4f %9.1f %9.1f\n", v.name.c_str(), med, mn, v.bytes / med / 1e6, v.bytes / mn / 1e6);
    }
    return 0;
}

__global__ void k_fill(uint64_t* p, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        p[i] = z ^ (z >> 31);
    }
}
