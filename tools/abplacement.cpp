// abplacement.cpp -- how much does WHERE a column lives matter?  The same fl_u32_unpack (W=7, 10 M blocks: 8.96 GB in, 40.96 GB
// out) on several independently hipMalloc'ed buffer pairs inside one process, every pair timed in interleaved rounds; then
// every input with every output (is it the input's or the output's placement?), and the pointers' low bits.
//   g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tools/abplacement.cpp -L fastlanes_amd -lfastlanes_amd \
//       -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../fastlanes_amd' -Wl,-rpath,/opt/rocm/lib -o tools/abplacement
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <hip/hip_runtime_api.h>
#include "fastlanes_amd.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv)
{
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
