// abplacement4.cpp -- abplacement2 for READS: two concurrent read-only streams (fl_u32_unpack_block_sums at W=32: 4 KiB read,
// 8 B written per block) over 4-GiB regions of one 200-GiB allocation, region 0 / 64 paired with every other region.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <hip/hip_runtime_api.h>
#include "fastlanes_amd.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main()
{
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
