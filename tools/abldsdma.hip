// abldsdma.hip -- read side of the wave-per-block kernels: input block -> LDS image through VGPRs (global load +
// ds_write_b128, the round-2 path) vs LDS-DMA (`buffer_load_dwordx4 ... lds`) with the default and the non-temporal
// cache policy.  Same buffers for every variant, interleaved rounds, median; outputs compared with the VGPR path first.
// Cases: BASELINE.json config 5 (u32, width[b] = 1 + b mod 32, 9 765 625 blocks) unpack and pack; uniform columns of the
// (T, W) VERDICT r02 names (u64 W=17, u32 W=7 / 12 / 20, u16 W=3 / 9, u8 W=3) -- unpack, pack, and the Delta / transpose
// pipeline kernel (k_chain) where its source stage is linear.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fastlanes_amd/csrc -I include tools/abldsdma.hip -o tools/abldsdma
// Run on the GPU box: tools/abldsdma [rounds] [quick]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#include "fl_widths.hpp"
#include "fl_chain.hpp"
#include "fl_scan.hpp"

using namespace fl;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_count_diff(const u32x4* a, const u32x4* b, uint64_t n_cells, unsigned long long* count)
{
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cells; i += (uint64_t)gridDim.x * blockDim.x) {
        const u32x4 x = a[i], y = b[i];
        bad += (x.x != y.x) | (x.y != y.y) | (x.z != y.z) | (x.w != y.w);
    }
    if (bad) atomicAdd(count, bad);
}

static char *g_in, *g_out, *g_ref, *g_aux;
static unsigned long long* g_count;
static int g_rounds = 5;

static uint64_t diff(const void* a, const void* b, uint64_t bytes)
{
    CK(hipMemset(g_count, 0, 8));
    hipLaunchKernelGGL(k_count_diff, dim3(8192), dim3(256), 0, 0, (const u32x4*)a, (const u32x4*)b, bytes / 16, g_count);
    unsigned long long h = 0;
    CK(hipMemcpy(&h, g_count, 8, hipMemcpyDeviceToHost));
    return h;
}

struct Variant { std::string name; std::function<void(char* out)> launch; std::vector<float> ms; };

// every variant writes g_out (timed) -- the first one (VGPR path) also wrote g_ref once for the comparison
static void run_case(const std::string& title, double bytes, uint64_t out_bytes, std::vector<Variant>& vs)
{
    vs[0].launch(g_ref);
    CK(hipDeviceSynchronize());
    std::string bad;
    for (size_t k = 1; k < vs.size(); ++k) {
        CK(hipMemset(g_out, 0xA5, out_bytes));
        vs[k].launch(g_out);
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        if (diff(g_ref, g_out, out_bytes & ~15ull) != 0) bad += " MISMATCH(" + vs[k].name + ")";
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& v : vs) v.launch(g_out);
    CK(hipDeviceSynchronize());
    for (int r = 0; r < g_rounds; ++r)
        for (auto& v : vs) {
            CK(hipEventRecord(e0, 0));
            v.launch(g_out);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            v.ms.push_back(ms);
        }
    CK(hipGetLastError());
    printf("%-34s%s\n", title.c_str(), bad.c_str());
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        printf("    %-28s %7.0f GB/s  (min-time %7.0f)\n", v.name.c_str(), bytes / v.ms[v.ms.size() / 2] / 1e6, bytes / v.ms[0] / 1e6);
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

static const char* RDN(int rd) { return rd == RD_VGPR ? "vgpr" : rd == RD_VGPR_NT ? "vgpr-nt" : rd == RD_DMA ? "dma" : rd == RD_AUTO ? "auto (shipped)" : "dma-nt"; }

template <typename T, bool PACK, int RD> void go_widths(WidthsArgs x, char* o, int wv)
{
    if (PACK) x.packed = o; else x.unpacked = o;
    (void)launch_widths<T, PACK, RD>(x, wv, 0);
}

template <typename T, bool PACK> void add_widths(std::vector<Variant>& vs, WidthsArgs a, std::initializer_list<int> waves)
{
    for (int wv : waves) {
        auto nm = [&](int rd) { return std::string(RDN(rd)) + ", " + std::to_string(wv) + " waves/SIMD"; };
        vs.push_back({nm(RD_VGPR), [=](char* o) { go_widths<T, PACK, RD_VGPR>(a, o, wv); }, {}});
        if (!PACK) vs.push_back({nm(RD_VGPR_NT), [=](char* o) { go_widths<T, PACK, RD_VGPR_NT>(a, o, wv); }, {}});   // pack's VGPR loads are nt already
        vs.push_back({nm(RD_DMA), [=](char* o) { go_widths<T, PACK, RD_DMA>(a, o, wv); }, {}});
        vs.push_back({nm(RD_DMA_NT), [=](char* o) { go_widths<T, PACK, RD_DMA_NT>(a, o, wv); }, {}});
        if (!PACK) vs.push_back({nm(RD_AUTO), [=](char* o) { go_widths<T, PACK, RD_AUTO>(a, o, wv); }, {}});
        {   // the shipped read path with workgroup b -> tile b instead of the XCD-contiguous map
            WidthsArgs lin = a;
            lin.linear_map = 1;
            vs.push_back({"shipped, LINEAR tile map, " + std::to_string(wv) + " w", [=](char* o) { go_widths<T, PACK, (PACK ? RD_VGPR : RD_AUTO)>(lin, o, wv); }, {}});
        }
    }
}

template <typename T> void uniform_case(unsigned W, bool pack, std::initializer_list<int> waves)
{
    constexpr unsigned TB = sizeof(T) * 8;
    const uint64_t bpb = 128ull * W + 128ull * TB;
    const uint64_t n = (12ull << 30) / bpb;
    // unpack: packed = g_in, unpacked = out;  pack: unpacked = g_in (full-entropy values, pack truncates), packed = out
    WidthsArgs a{pack ? nullptr : g_in, pack ? g_in : nullptr, nullptr, nullptr, nullptr, nullptr, 0, n, 0, W, 1, 0};
    std::vector<Variant> vs;
    if (pack) add_widths<T, true>(vs, a, waves); else add_widths<T, false>(vs, a, waves);
    char title[96];
    snprintf(title, sizeof title, "%s u%u W=%u (%llu blocks)", pack ? "pack" : "unpack", TB, W, (unsigned long long)n);
    run_case(title, (double)n * bpb, pack ? n * 128ull * W : n * 128ull * TB, vs);
}

template <typename T, int SRC, int BODY, int SNK> void chain_case(const char* name, unsigned W, std::initializer_list<int> waves)
{
    constexpr unsigned TB = sizeof(T) * 8;
    const bool pin = SRC == SRC_PACKED, pout = SNK == SNK_PACKED;
    const uint64_t bpb = (pin ? 128ull * W : 128ull * TB) + (pout ? 128ull * W : 128ull * TB) + (BODY != CHAIN_NONE ? 128 : 0);
    const uint64_t n = (12ull << 30) / bpb;
    ChainArgs a{};
    a.in = g_in;
    a.bases = g_aux;
    a.n_blocks = n;
    a.width = W;
    std::vector<Variant> vs;
    for (int wv : waves) {
        auto nm = [&](int rd) { return std::string(RDN(rd)) + ", " + std::to_string(wv) + " waves/SIMD"; };
        vs.push_back({nm(RD_VGPR), [=](char* o) { ChainArgs x = a; x.out = o; (void)launch_chain<T, SRC, BODY, SNK, RD_VGPR>(x, wv, 0); }, {}});
        if (SRC == SRC_PACKED) vs.push_back({nm(RD_VGPR_NT), [=](char* o) { ChainArgs x = a; x.out = o; (void)launch_chain<T, SRC, BODY, SNK, RD_VGPR_NT>(x, wv, 0); }, {}});
        vs.push_back({nm(RD_DMA), [=](char* o) { ChainArgs x = a; x.out = o; (void)launch_chain<T, SRC, BODY, SNK, RD_DMA>(x, wv, 0); }, {}});
        vs.push_back({nm(RD_DMA_NT), [=](char* o) { ChainArgs x = a; x.out = o; (void)launch_chain<T, SRC, BODY, SNK, RD_DMA_NT>(x, wv, 0); }, {}});
    }
    char title[96];
    snprintf(title, sizeof title, "%s u%u W=%u (%llu blocks)", name, TB, W, (unsigned long long)n);
    run_case(title, (double)n * bpb, pout ? n * 128ull * W : n * 128ull * TB, vs);
}

int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    g_rounds = argc > 1 ? atoi(argv[1]) : 5;
    const bool quick = argc > 2;
    const uint64_t cap = 13ull << 30;
    CK(hipMalloc(&g_in, 42ull << 30)); CK(hipMalloc(&g_out, 42ull << 30)); CK(hipMalloc(&g_ref, 42ull << 30)); CK(hipMalloc(&g_aux, 2ull << 30));
    CK(hipMalloc(&g_count, 8));
    CK(launch_fill_random((uint64_t*)g_in, (42ull << 30) / 8, 7, 0));
    CK(launch_fill_random((uint64_t*)g_aux, (2ull << 30) / 8, 8, 0));
    CK(hipDeviceSynchronize());
    (void)cap;
    printf("GB/s of algorithmic bytes, median of %d interleaved rounds on the same buffers\n", g_rounds);

    // ---- BASELINE.json config 5: u32, width[b] = 1 + b mod 32, 9 765 625 blocks --------------------------------------
    {
        const uint64_t n = 9765625ull;
        std::vector<uint8_t> hw(n);
        for (uint64_t b = 0; b < n; ++b) hw[b] = (uint8_t)(1 + b % 32);
        uint8_t* d_w; uint64_t *d_off, *d_total; uint32_t* d_err;
        CK(hipMalloc(&d_w, n)); CK(hipMalloc(&d_off, n * 8)); CK(hipMalloc(&d_total, 8)); CK(hipMalloc(&d_err, 4));
        CK(hipMemset(d_err, 0, 4));
        CK(hipMemcpy(d_w, hw.data(), n, hipMemcpyHostToDevice));
        ScanArgs sa{d_w, d_off, d_total, d_err, n, 32};
        CK(launch_widths_to_offsets(sa, 0));
        uint64_t pbytes = 0;
        CK(hipMemcpy(&pbytes, d_total, 8, hipMemcpyDeviceToHost));
        const double bytes = (double)pbytes + (double)n * 4096;
        {
            WidthsArgs a{g_in, nullptr, d_w, d_off, d_err, nullptr, 0, n, 0, 0, 1, pbytes};
            std::vector<Variant> vs;
            add_widths<uint32_t, false>(vs, a, {5, 6, 8});
            run_case("config 5: unpack u32 mixed 1..32", bytes, n * 4096, vs);
        }
        {
            WidthsArgs a{nullptr, g_in, d_w, d_off, d_err, nullptr, 0, n, 0, 0, 1, pbytes};
            std::vector<Variant> vs;
            add_widths<uint32_t, true>(vs, a, {6, 8});
            run_case("config 5: pack u32 mixed 1..32", bytes, pbytes, vs);
        }
        uint32_t herr = 0;
        CK(hipMemcpy(&herr, d_err, 4, hipMemcpyDeviceToHost));
        printf("    device err_flag %u\n", herr);
    }
    uniform_case<uint64_t>(17, false, {3, 4, 5});
    uniform_case<uint64_t>(17, true, {3, 4, 6});
    uniform_case<uint32_t>(7, true, {5, 6, 8});
    uniform_case<uint32_t>(7, false, {5, 6, 8});
    if (!quick) {
        uniform_case<uint32_t>(20, false, {4, 5, 6});
        uniform_case<uint32_t>(20, true, {4, 6});
        uniform_case<uint16_t>(3, true, {6, 8});
        uniform_case<uint16_t>(9, false, {6, 8});
        uniform_case<uint8_t>(6, false, {6, 8});
        uniform_case<uint64_t>(40, false, {3, 4});
    }
    chain_case<uint32_t, SRC_PACKED, CHAIN_UNDELTA, SNK_ROWS>("undelta_pack", 12, {4, 6});
    chain_case<uint16_t, SRC_PACKED, CHAIN_UNDELTA, SNK_ROWS>("undelta_pack", 9, {6, 8});
    chain_case<uint32_t, SRC_ROWS, CHAIN_UNDELTA, SNK_ROWS>("undelta", 32, {4, 6});
    chain_case<uint32_t, SRC_ROWS, CHAIN_NONE, SNK_ORIGINAL>("untranspose", 32, {4, 6});
    if (!quick) {
        chain_case<uint64_t, SRC_PACKED, CHAIN_UNDELTA, SNK_ORIGINAL>("undelta_pack_untranspose", 20, {3, 4});
        chain_case<uint16_t, SRC_ROWS, CHAIN_DELTA, SNK_ROWS>("delta", 16, {6, 8});
        chain_case<uint8_t, SRC_ORIGINAL, CHAIN_NONE, SNK_ROWS>("transpose", 8, {6, 8});
    }
    return 0;
}
