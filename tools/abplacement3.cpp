// abplacement3.cpp -- the codec kernel on buffers placed deliberately inside ONE big allocation whose 64-GiB parts behave as
// separate memory "zones" (tools/abplacement2): fl_u32_unpack W=7, 10 M blocks (8.96 GB in, 40.96 GB out).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <hip/hip_runtime_api.h>
#include "fastlanes_amd.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main()
{
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
