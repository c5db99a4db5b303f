// abbench.hip -- within-process interleaved A/B of launch-shape / cache-policy variants
// of the u32 W=7 unpack kernel, next to plain read / write / copy streams of the same
// byte mix (the in-situ ceilings).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17
// -I fastlanes_amd/csrc -I include tools/abbench.hip -L fastlanes_amd -lfastlanes_amd
// -Wl,-rpath,'$ORIGIN/../fastlanes_amd' -o tools/abbench ; run on the GPU box.
//   tools/abbench [n_blocks] [rounds]         round 1's u32 W=7 unpack experiments
//   tools/abbench pack64 [n_blocks] [rounds]  round 4: BASELINE config 3's pack leg (fl_u64_pack, W=17) against bare
//                                             64:17 read:write streams of exactly its bytes, same buffers, interleaved
//   tools/abbench thin <type bits> <W> [n_blocks] [rounds]   round 4: fl_<ty>_unpack_compare against a bare stream of its
//                                             bytes (W cells read : 1 cell written per thread), u16 / u8 W=3 and u32 W=7
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#include "fl_device.hpp"
#include "fastlanes_amd.h"
#include "fastlanes_amd_internal.h"

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

// ---------------------------------------------------------------------------------------------------------------
// mode "pack64" (round 4): is fl_u64_pack(17) at 10 M blocks (81.92 GB read, 21.76 GB written)
// at what the memory gives a bare stream of the same bytes at the same addresses?
// k_bare_wave has the product kernel's access shape and nothing else: one wavefront per 1024-value block, IN_G loads of
// 1 KiB contiguous (all in flight), OUT_G stores of 1 KiB contiguous through a descriptor that ends at the block's
// out_bytes (u64 W=17: 2 x 1 KiB + 128 B), XCD-contiguous tiles of 4 blocks, `sc1 nt` stores, occupancy set by the
// dynamic-LDS request -- no LDS traffic, no shifts.
// ---------------------------------------------------------------------------------------------------------------
template <int IN_G, int OUT_G, int LAUX>
__global__ __launch_bounds__(256) void k_bare_wave(const char* in, char* out, uint64_t n_blocks, uint64_t tiles_per_xcd, unsigned out_bytes)
{
    const uint64_t n_tiles = (n_blocks + 3) / 4;
    const uint64_t tile = (uint64_t)(blockIdx.x & 7u) * tiles_per_xcd + (blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint64_t blk = tile * 4 + wave;
    if (blk >= n_blocks) return;
    const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in) + blk * (uint64_t)(IN_G * 1024), 0, IN_G * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(out + blk * (uint64_t)out_bytes, 0, out_bytes, 0x00020000);
    u32x4 v[IN_G];
    static_for<IN_G>([&](auto G) { v[decltype(G)::value] = __builtin_amdgcn_raw_buffer_load_b128(in_rs, lane * 16u + decltype(G)::value * 1024u, 0, LAUX); });
    u32x4 acc = {0, 0, 0, 0};
    static_for<IN_G>([&](auto G) { acc ^= v[decltype(G)::value]; });
    static_for<OUT_G>([&](auto G) { __builtin_amdgcn_raw_buffer_store_b128(acc + (unsigned)decltype(G)::value, out_rs, lane * 16u + decltype(G)::value * 1024u, 0, 18); });
    if (OUT_G == 0 && acc.x == 0x12345678u && acc.y == 0x9abcdef0u) *reinterpret_cast<u32x4*>(out) = acc;   // keeps the loads of the read-only form alive
}

// the same bare stream under a WINDOWED tile map: the column is walked window by window (window_tiles tiles, a multiple of 8), and
// inside a window XCD x owns one contiguous eighth -- every XCD still sees one dense stream, but the whole chip reads from one
// stretch of the column at a time instead of from 8 places spread over all of it
template <int IN_G, int OUT_G, int LAUX>
__global__ __launch_bounds__(256) void k_bare_wave_windowed(const char* in, char* out, uint64_t n_blocks, uint64_t window_tiles, unsigned out_bytes)
{
    const uint64_t n_tiles = (n_blocks + 3) / 4;
    const uint64_t b = blockIdx.x;
    const uint64_t win = b / window_tiles, r = b - win * window_tiles;
    const uint64_t tile = win * window_tiles + (r & 7u) * (window_tiles >> 3) + (r >> 3);
    if (tile >= n_tiles) return;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint64_t blk = tile * 4 + wave;
    if (blk >= n_blocks) return;
    const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in) + blk * (uint64_t)(IN_G * 1024), 0, IN_G * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(out + blk * (uint64_t)out_bytes, 0, out_bytes, 0x00020000);
    u32x4 v[IN_G];
    static_for<IN_G>([&](auto G) { v[decltype(G)::value] = __builtin_amdgcn_raw_buffer_load_b128(in_rs, lane * 16u + decltype(G)::value * 1024u, 0, LAUX); });
    u32x4 acc = {0, 0, 0, 0};
    static_for<IN_G>([&](auto G) { acc ^= v[decltype(G)::value]; });
    static_for<OUT_G>([&](auto G) { __builtin_amdgcn_raw_buffer_store_b128(acc + (unsigned)decltype(G)::value, out_rs, lane * 16u + decltype(G)::value * 1024u, 0, 18); });
    if (OUT_G == 0 && acc.x == 0x12345678u && acc.y == 0x9abcdef0u) *reinterpret_cast<u32x4*>(out) = acc;
}

static unsigned lds_for_waves(int waves) { return (160u * 1024u / (unsigned)(waves < 3 ? 3 : waves)) & ~1023u; }

// ---------------------------------------------------------------------------------------------------------------
// mode "thin" (round 4): fl_<ty>_unpack_compare -- 128*W bytes read, 128 bytes of mask written per
// block -- against a bare stream of exactly those bytes in the kernel's own access shape: thread (block g of the wave,
// column c) loads W cells at a stride of 8 cells and stores ONE cell, the wave's 8 masks 1 KiB contiguous; XCD-contiguous
// tiles of 32 blocks, nt loads, `sc1 nt` store.  WR = 0: the same loads with nothing stored.
// ---------------------------------------------------------------------------------------------------------------
template <int RD, int WR, int MAXW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, MAXW)))
void k_bare_cells(const u32x4* in, u32x4* out, uint64_t n_blocks, uint64_t tiles_per_xcd)
{
    const uint64_t n_tiles = (n_blocks + 31) / 32;
    const uint64_t tile = (uint64_t)(blockIdx.x & 7u) * tiles_per_xcd + (blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    const unsigned tid = threadIdx.x;
    const uint64_t blk = tile * 32 + (tid >> 3);
    const unsigned c = tid & 7u;
    if (blk >= n_blocks) return;
    u32x4 v[RD];
    const u32x4* pk = in + blk * (uint64_t)(8 * RD) + c;
    static_for<RD>([&](auto I) { v[decltype(I)::value] = __builtin_nontemporal_load(pk + 8 * decltype(I)::value); });
    u32x4 acc = {0, 0, 0, 0};
    static_for<RD>([&](auto I) { acc ^= v[decltype(I)::value]; });
    if constexpr (WR > 0) {
        const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const uint64_t wave_first = tile * 32 + wave * 8u;
        const uint64_t left = n_blocks - wave_first;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(out) + wave_first * 128u, 0,
                                                                            (unsigned)(left < 8 ? left : 8) * 128u, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(acc, rs, ((tid >> 3) & 7u) * 128u + c * 16u, 0, 18);
    } else if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) {
        out[0] = acc;
    }
}

template <int W>
static int run_thin_w(unsigned type_bits, uint64_t n, int rounds)
{
    const uint64_t in_b = n * 128ull * W, out_b = n * 128ull;
    char *in, *out;
    CK(hipMalloc(&in, in_b));
    CK(hipMalloc(&out, out_b + 256));
    if (fl_fill_random(in, in_b, 42, nullptr) != 0) { printf("fl_fill_random failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    printf("# %s\n# unpack_compare u%u W=%d, %llu blocks: %.2f GB read + %.2f GB of mask written per launch (two hipMalloc's)\n", fl_version(),
           type_bits, W, (unsigned long long)n, in_b / 1e9, out_b / 1e9);
    std::vector<Variant> vs;
    const double bytes = (double)(in_b + out_b);
    auto cmp = [&](const char* name, int op) {
        vs.push_back({name, bytes, [=]() {
            int rc = type_bits == 16 ? fl_u16_unpack_compare(W, (const uint16_t*)in, op, (uint16_t)((1u << W) / 2), n, (uint32_t*)out, nullptr)
                   : type_bits == 8  ? fl_u8_unpack_compare(W, (const uint8_t*)in, op, (uint8_t)((1u << W) / 2), n, (uint32_t*)out, nullptr)
                                     : fl_u32_unpack_compare(W, (const uint32_t*)in, op, (1u << W) / 2, n, (uint32_t*)out, nullptr);
            if (rc != 0) { printf("unpack_compare failed %d\n", rc); exit(1); }
        }, {}});
    };
    cmp("fl_unpack_compare  x < k", FL_CMP_LT);
    cmp("fl_unpack_compare  x == k", FL_CMP_EQ);
    for (int wv : {3, 4, 5, 6}) {
        static char names[8][64];
        snprintf(names[wv], 64, "fl_unpack_compare  x < k, launched at %d waves/SIMD", wv);
        vs.push_back({names[wv], bytes, [=]() {
            fl_internal_set_kernel_policy(2 + 256 * wv);
            int rc = type_bits == 16 ? fl_u16_unpack_compare(W, (const uint16_t*)in, FL_CMP_LT, (uint16_t)((1u << W) / 2), n, (uint32_t*)out, nullptr)
                   : type_bits == 8  ? fl_u8_unpack_compare(W, (const uint8_t*)in, FL_CMP_LT, (uint8_t)((1u << W) / 2), n, (uint32_t*)out, nullptr)
                                     : fl_u32_unpack_compare(W, (const uint32_t*)in, FL_CMP_LT, (1u << W) / 2, n, (uint32_t*)out, nullptr);
            fl_internal_set_kernel_policy(0);
            if (rc != 0) { printf("unpack_compare failed %d\n", rc); exit(1); }
        }, {}});
    }
    vs.push_back({"fl_unpack_block_sums (8 B written per block)", (double)(in_b + n * 8), [=]() {
        int rc = type_bits == 16 ? fl_u16_unpack_block_sums(W, (const uint16_t*)in, n, (uint64_t*)out, nullptr)
               : type_bits == 8  ? fl_u8_unpack_block_sums(W, (const uint8_t*)in, n, (uint64_t*)out, nullptr)
                                 : fl_u32_unpack_block_sums(W, (const uint32_t*)in, n, (uint64_t*)out, nullptr);
        if (rc != 0) { printf("unpack_block_sums failed %d\n", rc); exit(1); }
    }, {}});
    const uint64_t tpx = ((n + 31) / 32 + 7) / 8;
    auto bare = [&](const char* name, auto kern, double by) {
        vs.push_back({name, by, [=]() { hipLaunchKernelGGL(kern, dim3((unsigned)(tpx * 8)), dim3(256), 0, 0, (const u32x4*)in, (u32x4*)out, n, tpx); }, {}});
    };
    bare("bare stream W cells rd : 1 cell wr, 8 waves", (k_bare_cells<W, 1, 8>), bytes);
    bare("bare stream W cells rd : 1 cell wr, 4 waves", (k_bare_cells<W, 1, 4>), bytes);
    bare("bare stream W cells rd : 1 cell wr, 2 waves", (k_bare_cells<W, 1, 2>), bytes);
    bare("bare stream W cells rd, nothing stored, 8 waves", (k_bare_cells<W, 0, 8>), (double)in_b);
    cmp("fl_unpack_compare  x < k (again)", FL_CMP_LT);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& v : vs) v.launch();
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
    printf("%-56s %9s %9s %9s %9s %7s\n", "variant", "med_ms", "min_ms", "GB/s_med", "GB/s_max", "frac");
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        float med = v.ms[v.ms.size() / 2], mn = v.ms[0];
        printf("%-56s %9.4f %9.4f %9.1f %9.1f %7.3f\n", v.name.c_str(), med, mn, v.bytes / med / 1e6, v.bytes / mn / 1e6, v.bytes / med / 1e6 / 8000.0);
    }
    CK(hipFree(in)); CK(hipFree(out));
    return 0;
}

static int run_thin(int argc, char** argv)
{
    const unsigned type_bits = argc > 2 ? (unsigned)atoi(argv[2]) : 16;
    const int w = argc > 3 ? atoi(argv[3]) : 3;
    const uint64_t n = argc > 4 ? strtoull(argv[4], 0, 10) : 40000000ull;
    const int rounds = argc > 5 ? atoi(argv[5]) : 7;
    if (w == 3 && (type_bits == 16 || type_bits == 8)) return run_thin_w<3>(type_bits, n, rounds);
    if (w == 7 && type_bits == 32) return run_thin_w<7>(type_bits, n, rounds);
    printf("thin: built for u16 / u8 W=3 and u32 W=7\n");
    return 2;
}

static int run_pack64(int argc, char** argv)
{
    const uint64_t n = argc > 2 ? strtoull(argv[2], 0, 10) : 10000000ull;
    const int rounds = argc > 3 ? atoi(argv[3]) : 5;
    const unsigned W = 17;
    const uint64_t in_b = n * 8192ull, out_b = n * 128ull * W;
    char *in, *out;
    CK(hipMalloc(&in, in_b));
    CK(hipMalloc(&out, out_b));
    if (fl_fill_random(in, in_b, 42, nullptr) != 0) { printf("fl_fill_random failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    printf("# %s\n# pack u64 W=17, %llu blocks: %.2f GB read + %.2f GB written per launch; in %p out %p (two hipMalloc's)\n", fl_version(),
           (unsigned long long)n, in_b / 1e9, out_b / 1e9, (void*)in, (void*)out);
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    std::vector<Variant> vs;
    const double bytes = (double)(in_b + out_b);
    auto lib = [&](const char* name, int policy, uint64_t first, uint64_t count) {
        vs.push_back({name, (double)count * (8192 + 128 * W), [=]() {
            fl_internal_set_kernel_policy(policy);
            if (fl_u64_pack(W, (const uint64_t*)(in + first * 8192), (uint64_t*)(out + first * 128 * W), count, nullptr) != 0) { printf("fl_u64_pack failed\n"); exit(1); }
            fl_internal_set_kernel_policy(0);
        }, {}});
    };
    lib("fl_u64_pack(17) as shipped", 0, 0, n);
    for (int w : {3, 4, 5, 6, 8}) {
        static char names[9][64];
        snprintf(names[w], 64, "fl_u64_pack(17) wave-per-block, %d waves/SIMD", w);
        lib(names[w], 2 + 256 * w, 0, n);
    }
    // size / position dependence inside the same buffers
    lib("fl_u64_pack(17) first half only", 0, 0, n / 2);
    lib("fl_u64_pack(17) second half only", 0, n / 2, n - n / 2);
    lib("fl_u64_pack(17) first quarter only", 0, 0, n / 4);
    lib("fl_u64_pack(17) last quarter only", 0, n - n / 4, n / 4);
    // the two halves at once, on two streams (each input inside fewer memory classes)
    hipEvent_t fork, join;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    vs.push_back({"fl_u64_pack(17) two half-column launches, two streams", bytes, [=]() {
        CK(hipEventRecord(fork, 0));
        CK(hipStreamWaitEvent(s1, fork, 0));
        CK(hipStreamWaitEvent(s2, fork, 0));
        fl_u64_pack(W, (const uint64_t*)in, (uint64_t*)out, n / 2, s1);
        fl_u64_pack(W, (const uint64_t*)(in + (n / 2) * 8192), (uint64_t*)(out + (n / 2) * 128 * W), n - n / 2, s2);
        CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(0, join, 0));
        CK(hipEventRecord(join, s2)); CK(hipStreamWaitEvent(0, join, 0));
    }, {}});
    const uint64_t tpx = ((n + 3) / 4 + 7) / 8;
    auto bare = [&](const char* name, auto kern, int waves) {
        const unsigned lds = lds_for_waves(waves);
        vs.push_back({name, bytes, [=]() { hipLaunchKernelGGL(kern, dim3((unsigned)(tpx * 8)), dim3(256), lds, 0, (const char*)in, out, n, tpx, 128u * W); }, {}});
    };
    bare("bare wave stream 8 KiB rd : 2176 B wr, nt loads, 3 waves", (k_bare_wave<8, 3, 2>), 3);
    bare("bare wave stream 8 KiB rd : 2176 B wr, nt loads, 4 waves", (k_bare_wave<8, 3, 2>), 4);
    bare("bare wave stream 8 KiB rd : 2176 B wr, nt loads, 5 waves", (k_bare_wave<8, 3, 2>), 5);
    bare("bare wave stream 8 KiB rd : 2176 B wr, nt loads, 6 waves", (k_bare_wave<8, 3, 2>), 6);
    bare("bare wave stream 8 KiB rd : 2176 B wr, nt loads, 8 waves", (k_bare_wave<8, 3, 2>), 8);
    bare("bare wave stream 8 KiB rd : 2176 B wr, default loads, 5 waves", (k_bare_wave<8, 3, 0>), 5);
    bare("bare wave stream 8 KiB rd : 2176 B wr, default loads, 8 waves", (k_bare_wave<8, 3, 0>), 8);
    {
        // round 1's tile-shaped bare stream (256 threads x 64 cells read : 17 written)
        const uint64_t n_tiles = n / 32, tpx32 = (n_tiles + 7) / 8;
        u32x4* sink = nullptr;
        vs.push_back({"tuned tile stream 64rd:17wr maxw1", (double)n_tiles * 32 * (8192 + 128 * W), [=]() {
            hipLaunchKernelGGL((k_stream_tuned<64, 17, 1>), dim3((unsigned)(tpx32 * 8)), dim3(256), 0, 0, (const u32x4*)in, (u32x4*)out, n_tiles, tpx32, sink); }, {}});
        vs.push_back({"tuned tile stream 64rd:17wr maxw2", (double)n_tiles * 32 * (8192 + 128 * W), [=]() {
            hipLaunchKernelGGL((k_stream_tuned<64, 17, 2>), dim3((unsigned)(tpx32 * 8)), dim3(256), 0, 0, (const u32x4*)in, (u32x4*)out, n_tiles, tpx32, sink); }, {}});
        vs.push_back({"read-only bare wave stream 8 KiB (no stores)", (double)in_b, [=]() {
            hipLaunchKernelGGL((k_bare_wave<8, 0, 2>), dim3((unsigned)(tpx * 8)), dim3(256), lds_for_waves(5), 0, (const char*)in, out, n, tpx, 128u * W); }, {}});
    }
    {
        // windowed tile map (see k_bare_wave_windowed): window = 2^k tiles of 4 blocks (32 KiB of input each)
        static char names[2][8][96];
        int i = 0;
        for (int k : {11, 14, 17, 19, 21}) {
            const uint64_t wt = 1ull << k;
            const uint64_t grid = (((n + 3) / 4 + wt - 1) / wt) * wt;
            snprintf(names[0][i], 96, "bare wave stream, nt, 5 waves, WINDOWED map: %6.0f MiB of input per window", wt * 32768.0 / (1 << 20));
            vs.push_back({names[0][i], bytes, [=]() {
                hipLaunchKernelGGL((k_bare_wave_windowed<8, 3, 2>), dim3((unsigned)grid), dim3(256), lds_for_waves(5), 0, (const char*)in, out, n, wt, 128u * W); }, {}});
            snprintf(names[1][i], 96, "read-only bare wave stream, WINDOWED map: %6.0f MiB per window", wt * 32768.0 / (1 << 20));
            vs.push_back({names[1][i], (double)in_b, [=]() {
                hipLaunchKernelGGL((k_bare_wave_windowed<8, 0, 2>), dim3((unsigned)grid), dim3(256), lds_for_waves(5), 0, (const char*)in, out, n, wt, 128u * W); }, {}});
            ++i;
        }
    }
    lib("fl_u64_pack(17) as shipped (again)", 0, 0, n);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& v : vs) v.launch();
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
    printf("%-66s %9s %9s %9s %9s %7s\n", "variant", "med_ms", "min_ms", "GB/s_med", "GB/s_max", "frac");
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        float med = v.ms[v.ms.size() / 2], mn = v.ms[0];
        printf("%-66s %9.4f %9.4f %9.1f %9.1f %7.3f\n", v.name.c_str(), med, mn, v.bytes / med / 1e6, v.bytes / mn / 1e6, v.bytes / med / 1e6 / 8000.0);
    }
    return 0;
}

int main(int argc, char** argv)
{
    if (argc > 1 && std::string(argv[1]) == "pack64") return run_pack64(argc, argv);
    if (argc > 1 && std::string(argv[1]) == "thin") return run_thin(argc, argv);
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
