// abplacement.cpp -- round 3's four placement probes in one file (they were abplacement.cpp, abplacement2.cpp, abplacement3.cpp and
// abplacement4.cpp until round 6; their results are profiles/abplacement_r03.txt and profiles/abzones_r03.txt):
//   abplacement pairs [...]        the same fl_u32_unpack on several independently hipMalloc'ed buffer pairs, every input with every output
//   abplacement write-zones [...]  two concurrent 4-GiB WRITE streams over every pairing of regions of a 200-GiB allocation
//   abplacement zones              the codec kernel with its buffers placed deliberately inside one big allocation
//   abplacement read-zones         write-zones for READS
//   g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tools/abplacement.cpp -L fastlanes_amd -lfastlanes_amd \
//       -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../fastlanes_amd' -Wl,-rpath,/opt/rocm/lib -o tools/abplacement
// (Round 6 superseded the "zones" reading of these results: tools/exp_vmm.cpp, profiles/r06_vmm_placement.txt.)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <hip/hip_runtime_api.h>
#include "fastlanes_amd.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- mode pairs (was abplacement.cpp) --------------------------------------------------
// abplacement.cpp -- how much does WHERE a column lives matter?  The same fl_u32_unpack (W=7, 10 M blocks: 8.96 GB in, 40.96 GB
// out) on several independently hipMalloc'ed buffer pairs inside one process, every pair timed in interleaved rounds; then
// every input with every output (is it the input's or the output's placement?), and the pointers' low bits.
static int run_pairs(int argc, char** argv)
{
    (void)argc; (void)argv;
    const size_t n = 10000000, ib = n * 896, ob = n * 4096;
    const int P = argc > 1 ? atoi(argv[1]) : 4, rounds = argc > 2 ? atoi(argv[2]) : 5;
    std::vector<uint32_t*> in(P), out(P);
    for (int p = 0; p < P; ++p) {
        CK(hipMalloc((void**)&in[p], ib));
        CK(hipMalloc((void**)&out[p], ob));
        if (fl_fill_random(in[p], ib, 100 + p, nullptr) != FL_OK) return 1;
        printf("pair %d: in %p (mod 2MiB %#zx, mod 1GiB %#zx)  out %p (mod 2MiB %#zx, mod 1GiB %#zx)\n", p, (void*)in[p],
               (size_t)in[p] & ((2u << 20) - 1), (size_t)in[p] & ((1u << 30) - 1), (void*)out[p], (size_t)out[p] & ((2u << 20) - 1), (size_t)out[p] & ((1u << 30) - 1));
    }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_pair = [&](int i, int o) {
        std::vector<float> ms;
        for (int r = 0; r < rounds; ++r) {
            CK(hipEventRecord(e0, nullptr));
            if (fl_u32_unpack(7, in[i], out[o], n, nullptr) != FL_OK) exit(1);
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        return (double)(ib + ob) / ms[ms.size() / 2] / 1e6;
    };
    for (int p = 0; p < P; ++p) (void)time_pair(p, p);   // warm
    printf("GB/s, median of %d: rows = input buffer, columns = output buffer\n        ", rounds);
    for (int o = 0; o < P; ++o) printf("  out%d ", o);
    printf("\n");
    for (int i = 0; i < P; ++i) {
        printf("in%d    ", i);
        for (int o = 0; o < P; ++o) printf(" %6.0f", time_pair(i, o));
        printf("\n");
    }
    // is a slow output buffer slow everywhere, or in places?  8 consecutive eighths of every pair, each timed on its own
    printf("GB/s per eighth of the column (1.25 M blocks each), pair by pair\n");
    const size_t n8 = n / 8;
    for (int p = 0; p < P; ++p) {
        printf("pair %d  ", p);
        for (int k = 0; k < 8; ++k) {
            std::vector<float> ms;
            for (int r = 0; r < rounds; ++r) {
                CK(hipEventRecord(e0, nullptr));
                if (fl_u32_unpack(7, in[p] + k * n8 * 224, out[p] + k * n8 * 1024, n8, nullptr) != FL_OK) exit(1);
                CK(hipEventRecord(e1, nullptr));
                CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1));
                ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            printf(" %6.0f", (double)n8 * 4992 / ms[ms.size() / 2] / 1e6);
        }
        printf("\n");
    }
    // the same window (n/8 blocks) slid over the column in steps of n/32: where in the allocation is it fast?
    printf("GB/s of a 1.25 M-block window starting at block s = j * 312 500, pairs 0 and 1\n");
    for (int p = 0; p < 2 && p < P; ++p) {
        printf("pair %d ", p);
        for (int j = 0; j <= 28; ++j) {
            const size_t s0 = (size_t)j * (n / 32);
            std::vector<float> ms;
            for (int r = 0; r < rounds; ++r) {
                CK(hipEventRecord(e0, nullptr));
                if (fl_u32_unpack(7, in[p] + s0 * 224, out[p] + s0 * 1024, n8, nullptr) != FL_OK) exit(1);
                CK(hipEventRecord(e1, nullptr));
                CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1));
                ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            printf(" %5.0f", (double)n8 * 4992 / ms[ms.size() / 2] / 1e6);
        }
        printf("\n");
    }
    // is it the kernel or the memory?  a plain write-only stream (fl_fill_random) over the same output windows
    printf("write-only GB/s (fl_fill_random) over the same output windows, pairs 0 and 1\n");
    for (int p = 0; p < 2 && p < P; ++p) {
        printf("pair %d ", p);
        for (int j = 0; j <= 28; ++j) {
            const size_t s0 = (size_t)j * (n / 32);
            std::vector<float> ms;
            for (int r = 0; r < rounds; ++r) {
                CK(hipEventRecord(e0, nullptr));
                if (fl_fill_random(out[p] + s0 * 1024, n8 * 4096, 7, nullptr) != FL_OK) exit(1);
                CK(hipEventRecord(e1, nullptr));
                CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1));
                ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            printf(" %5.0f", (double)n8 * 4096 / ms[ms.size() / 2] / 1e6);
        }
        printf("\n");
    }
    return 0;
}

// ---- mode write-zones (was abplacement2.cpp) --------------------------------------------------
// abplacement2.cpp -- follow-up to abplacement: which parts of a big allocation can be written concurrently at more than the
// single-stream rate?  Two concurrent 4-GiB write streams (fl_fill_random on two hipStreams): region 0 paired with every
// other 4-GiB region of a 200-GiB allocation, then a few other anchors.
static int run_write_zones(int argc, char** argv)
{
    (void)argc; (void)argv;
    const size_t GiB = 1ull << 30, TOT = argc > 1 ? (size_t)atoi(argv[1]) : 200, R = 4;
    char* big;
    CK(hipMalloc((void**)&big, TOT * GiB));
    printf("big %p (%zu GiB)\n", (void*)big, TOT);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    auto two = [&](size_t p, size_t q) {
        std::vector<float> ms;
        for (int r = 0; r < 3; ++r) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, nullptr));
            fl_fill_random(big + p * GiB, R * GiB, 1, s1);
            fl_fill_random(big + q * GiB, R * GiB, 2, s2);
            CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        return 2.0 * R * GiB / ms[1] / 1e6;
    };
    for (size_t anchor : {(size_t)0, (size_t)64, (size_t)100}) {
        if (anchor + R > TOT) continue;
        printf("anchor [%zu,%zu) GiB paired with [q,q+4), q = 0,4,8,...: GB/s of both\n", anchor, anchor + R);
        for (size_t q = 0; q + R <= TOT; q += R) {
            if (q == anchor) { printf("     -"); } else printf(" %5.0f", two(anchor, q));
            if ((q / R) % 16 == 15) printf("\n");
        }
        printf("\n");
    }
    return 0;
}

// ---- mode zones (was abplacement3.cpp) --------------------------------------------------
// abplacement3.cpp -- the codec kernel on buffers placed deliberately inside ONE big allocation whose 64-GiB parts behave as
// separate memory "zones" (tools/abplacement2): fl_u32_unpack W=7, 10 M blocks (8.96 GB in, 40.96 GB out).
static int run_zones(int argc, char** argv)
{
    (void)argc; (void)argv;
    const size_t GiB = 1ull << 30, n = 10000000, ib = n * 896, ob = n * 4096;
    char* big;
    CK(hipMalloc((void**)&big, 200 * GiB));
    if (fl_fill_random(big, 200 * GiB, 3, nullptr) != FL_OK) return 1;
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto rate = [&](double in_gib, double out_gib, int w) {
        const uint32_t* in = (const uint32_t*)(big + (size_t)(in_gib * GiB));
        uint32_t* out = (uint32_t*)(big + (size_t)(out_gib * GiB));
        const size_t pb = n * 128ull * w;
        std::vector<float> ms;
        for (int r = 0; r < 6; ++r) {
            CK(hipEventRecord(e0, nullptr));
            if (fl_u32_unpack(w, in, out, n, nullptr) != FL_OK) exit(1);
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (r) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        return (double)(pb + ob) / ms[2] / 1e6;
    };
    (void)ib;
    printf("fl_u32_unpack W=7, 10 M blocks; offsets in GiB inside a 200-GiB allocation (zones = 64-GiB parts)\n");
    printf("  in [0,8.4)    out [10,48.2)   same zone                         %6.0f GB/s\n", rate(0, 10, 7));
    printf("  in [0,8.4)    out [64,102.2)  in zone 0, out zone 1             %6.0f\n", rate(0, 64, 7));
    printf("  in [0,8.4)    out [45,83.2)   out straddles 64 (half / half)    %6.0f\n", rate(0, 45, 7));
    printf("  in [60,68.4)  out [109,147.2) both straddle (64 and 128)        %6.0f\n", rate(60, 109, 7));
    printf("  in [120,128.4) out [45,83.2)  in zone 1/2, out straddles 64     %6.0f\n", rate(120, 45, 7));
    printf("  in [150,158.4) out [45,83.2)  in zone 2, out straddles 64       %6.0f\n", rate(150, 45, 7));
    printf("  in [0,8.4)    out [26,64.2)   out ends at the boundary          %6.0f\n", rate(0, 26, 7));
    printf("u32 W=12 (in 14.3 GiB): same zone %6.0f | out straddles %6.0f | in zone 2, out straddles %6.0f\n", rate(0, 16, 12), rate(0, 45, 12), rate(150, 45, 12));
    printf("u32 W=20 (in 23.8 GiB): same zone %6.0f | out straddles %6.0f | in zone 2, out straddles %6.0f\n", rate(0, 24.5, 20), rate(0, 45, 20), rate(150, 45, 20));
    return 0;
}

// ---- mode read-zones (was abplacement4.cpp) --------------------------------------------------
// abplacement4.cpp -- abplacement2 for READS: two concurrent read-only streams (fl_u32_unpack_block_sums at W=32: 4 KiB read,
// 8 B written per block) over 4-GiB regions of one 200-GiB allocation, region 0 / 64 paired with every other region.
static int run_read_zones(int argc, char** argv)
{
    (void)argc; (void)argv;
    const size_t GiB = 1ull << 30, TOT = 200, R = 4, nblk = R * GiB / 4096;
    char* big; uint64_t *s1o, *s2o;
    CK(hipMalloc((void**)&big, TOT * GiB));
    CK(hipMalloc((void**)&s1o, nblk * 8)); CK(hipMalloc((void**)&s2o, nblk * 8));
    fl_fill_random(big, TOT * GiB, 3, nullptr);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    auto two = [&](size_t p, size_t q, bool both) {
        std::vector<float> ms;
        for (int r = 0; r < 3; ++r) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, nullptr));
            fl_u32_unpack_block_sums(32, (const uint32_t*)(big + p * GiB), nblk, s1o, s1);
            if (both) fl_u32_unpack_block_sums(32, (const uint32_t*)(big + q * GiB), nblk, s2o, s2);
            CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        return (both ? 2.0 : 1.0) * R * GiB / ms[1] / 1e6;
    };
    printf("one read stream over [0,4) GiB: %6.0f GB/s\n", two(0, 0, false));
    for (size_t anchor : {(size_t)0, (size_t)64}) {
        printf("READ anchor [%zu,%zu) GiB paired with [q,q+4), q = 0,4,8,...: GB/s of both\n", anchor, anchor + R);
        for (size_t q = 0; q + R <= TOT; q += R) {
            if (q == anchor) printf("     -"); else printf(" %5.0f", two(anchor, q, true));
            if ((q / R) % 16 == 15) printf("\n");
        }
        printf("\n");
    }
    return 0;
}

int main(int argc, char** argv)
{
    const char* mode = argc > 1 ? argv[1] : "pairs";
    // every mode sees its own arguments where the stand-alone tool saw them: argv[1], argv[2], ...
    if (!strcmp(mode, "pairs")) return run_pairs(argc - 1, argv + 1);
    if (!strcmp(mode, "write-zones")) return run_write_zones(argc - 1, argv + 1);
    if (!strcmp(mode, "zones")) return run_zones(argc - 1, argv + 1);
    if (!strcmp(mode, "read-zones")) return run_read_zones(argc - 1, argv + 1);
    printf("usage: abplacement pairs | write-zones | zones | read-zones\n");
    return 2;
}
