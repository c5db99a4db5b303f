// abuniform.hip -- uniform-width columns: the shipped per-(T,W) cell-column kernels (fl_<ty>_pack / _unpack) vs the
// generic wave-per-block kernels of fl_widths.hpp run with one width for every block, interleaved in one process on the
// same buffers.  Build like tools/abmixed.hip (against libfastlanes_amd_full.so); run on the GPU box: tools/abuniform [rounds] [width stride] [GiB per launch] [blocks per wavefront, prefetched; 0 = library default] [types, e.g. u8u16]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#include "fastlanes_amd.h"
#include "fastlanes_amd_internal.h"
#include "fl_widths.hpp"

using namespace fl;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_fill(uint64_t* p, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        p[i] = z ^ (z >> 31);
    }
}
__global__ void k_count_diff(const u32x4* a, const u32x4* b, uint64_t n_cells, unsigned long long* count)
{
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cells; i += (uint64_t)gridDim.x * blockDim.x) {
        const u32x4 x = a[i], y = b[i];
        bad += (x.x != y.x) | (x.y != y.y) | (x.z != y.z) | (x.w != y.w);
    }
    if (bad) atomicAdd(count, bad);
}

template <typename T, bool PACK> void launch_w(const WidthsArgs& a, int waves) { (void)launch_widths<T, PACK>(a, waves, 0); }

template <typename T> struct Abi;
template <> struct Abi<uint8_t> { static constexpr auto pack = fl_u8_pack; static constexpr auto unpack = fl_u8_unpack; static constexpr const char* name = "u8"; };
template <> struct Abi<uint16_t> { static constexpr auto pack = fl_u16_pack; static constexpr auto unpack = fl_u16_unpack; static constexpr const char* name = "u16"; };
template <> struct Abi<uint32_t> { static constexpr auto pack = fl_u32_pack; static constexpr auto unpack = fl_u32_unpack; static constexpr const char* name = "u32"; };
template <> struct Abi<uint64_t> { static constexpr auto pack = fl_u64_pack; static constexpr auto unpack = fl_u64_unpack; static constexpr const char* name = "u64"; };

struct Variant { std::string name; std::function<void()> launch; std::vector<float> ms; };

static char *g_un, *g_pk, *g_un2, *g_pk2;
static unsigned long long* g_count;
static unsigned g_bpw = 0;       // 0 = what the library uses for uniform-width calls (fl_dispatch.hpp: uniform_blocks_per_wave); else forced, prefetched
static uint64_t g_gb = 16;       // bytes moved per launch (GiB): BASELINE's configs are 50-100 GB columns, and the best occupancy
                                 // of a few (T, W) moves with the column size (u64 W=15..17: 3 waves at 16 GB, 4+ at 100 GB)

static uint64_t diff(const void* a, const void* b, uint64_t bytes)
{
    CK(hipMemset(g_count, 0, 8));
    hipLaunchKernelGGL(k_count_diff, dim3(8192), dim3(256), 0, 0, (const u32x4*)a, (const u32x4*)b, bytes / 16, g_count);
    unsigned long long h = 0;
    CK(hipMemcpy(&h, g_count, 8, hipMemcpyDeviceToHost));
    return h;
}

struct Row { unsigned tb, w; double cc_u, cc_p, wu[5], wp[5]; };
static const int WAVES[5] = {3, 4, 5, 6, 8};
static std::vector<Row> g_rows;

template <typename T> void run(unsigned W, int rounds)
{
    constexpr unsigned TB = sizeof(T) * 8;
    const uint64_t bpb = 128ull * W + 128ull * TB;
    const uint64_t n = (g_gb << 30) / bpb;
    const double bytes = (double)n * bpb;
    T* un = (T*)g_un; T* pk = (T*)g_pk; T* pk2 = (T*)g_pk2;
    const unsigned bpw_u = g_bpw ? g_bpw : uniform_blocks_per_wave(TB, false), bpw_p = g_bpw ? g_bpw : uniform_blocks_per_wave(TB, true);
    WidthsArgs up{g_pk, g_un2, nullptr, nullptr, nullptr, nullptr, 0, n, 0, W, bpw_u, 0, bpw_u > 1};
    WidthsArgs pa{g_pk2, g_un, nullptr, nullptr, nullptr, nullptr, 0, n, 0, W, bpw_p, 0, bpw_p > 1};
    // correctness on these very buffers (library forced onto its cell-column kernels: policy 1)
    fl_internal_set_kernel_policy(1);
    Abi<T>::unpack(W, pk, un, n, nullptr);
    launch_w<T, false>(up, 6);
    CK(hipDeviceSynchronize());
    const bool ok_u = diff(g_un, g_un2, n * 128ull * TB) == 0;
    up.unpacked = g_un;     // timed runs: every variant reads and writes the SAME buffers (placement moves results by +-5 %)
    Abi<T>::pack(W, un, pk, n, nullptr);
    launch_w<T, true>(pa, 4);
    CK(hipDeviceSynchronize());
    const bool ok_p = W == 0 || diff(g_pk, g_pk2, n * 128ull * W) == 0;
    // timed pack runs read FULL-ENTROPY values (g_un2, refilled with random bits; pack truncates them) -- W-bit values
    // are mostly zero bits and would let DVFS inflate the clocks; nothing writes g_un2 during the timed rounds
    hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, 0, (uint64_t*)g_un2, (n * 128ull * TB) / 8);
    CK(hipDeviceSynchronize());
    pa.unpacked = g_un2;
    T* unr = (T*)g_un2;
    std::vector<Variant> vs;
    vs.push_back({"unpack cc", [=] { Abi<T>::unpack(W, pk, un, n, nullptr); }, {}});
    for (int k = 0; k < 5; ++k) vs.push_back({"unpack wpb", [=] { launch_w<T, false>(up, WAVES[k]); }, {}});
    vs.push_back({"pack cc", [=] { Abi<T>::pack(W, unr, pk2, n, nullptr); }, {}});
    for (int k = 0; k < 5; ++k) vs.push_back({"pack wpb", [=] { launch_w<T, true>(pa, WAVES[k]); }, {}});
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
    Row row{TB, W, 0, 0, {0}, {0}};
    auto gbps = [&](Variant& v) { std::sort(v.ms.begin(), v.ms.end()); return bytes / v.ms[v.ms.size() / 2] / 1e6; };
    row.cc_u = gbps(vs[0]);
    for (int k = 0; k < 5; ++k) row.wu[k] = gbps(vs[1 + k]);
    row.cc_p = gbps(vs[6]);
    for (int k = 0; k < 5; ++k) row.wp[k] = gbps(vs[7 + k]);
    g_rows.push_back(row);
    printf("u%-2u W=%-2u %s%s| unpack cc %6.0f  wpb", TB, W, ok_u ? "" : "UNPACK-MISMATCH ", ok_p ? "" : "PACK-MISMATCH ", row.cc_u);
    for (int k = 0; k < 5; ++k) printf(" %6.0f", row.wu[k]);
    printf(" | pack cc %6.0f  wpb", row.cc_p);
    for (int k = 0; k < 5; ++k) printf(" %6.0f", row.wp[k]);
    printf("\n");
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

template <typename T> void run_all(int rounds, int stride)
{
    for (unsigned w = 0; w <= sizeof(T) * 8; w += (w < 8 || w + (unsigned)stride > sizeof(T) * 8 - 2) ? 1 : stride) run<T>(w, rounds);
}

int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int rounds = argc > 1 ? atoi(argv[1]) : 5;
    if (argc > 3) g_gb = strtoull(argv[3], nullptr, 10);
    if (argc > 4) g_bpw = (unsigned)atoi(argv[4]);
    const uint64_t cap = ((g_gb + 1) << 30);
    CK(hipMalloc(&g_un, cap)); CK(hipMalloc(&g_pk, cap)); CK(hipMalloc(&g_un2, cap)); CK(hipMalloc(&g_pk2, cap));
    CK(hipMalloc(&g_count, 8));
    hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, 0, (uint64_t*)g_pk, cap / 8);
    hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, 0, (uint64_t*)g_un, cap / 8);
    CK(hipDeviceSynchronize());
    const int stride = argc > 2 ? atoi(argv[2]) : 1;
    printf("GB/s (algorithmic bytes), median of %d, %llu GiB per launch; cc = cell-column kernel, wpb = wave-per-block at 3 4 5 6 8 waves/SIMD\n", rounds, (unsigned long long)g_gb);
    const char* only = argc > 5 ? argv[5] : "";      // e.g. "u8u16"
    if (!*only || strstr(only, "u32")) run_all<uint32_t>(rounds, stride);
    if (!*only || strstr(only, "u64")) run_all<uint64_t>(rounds, stride);
    if (!*only || strstr(only, "u16")) run_all<uint16_t>(rounds, stride);
    if (!*only || strstr(only, "u8")) run_all<uint8_t>(rounds, stride);
    return 0;
}
