// abplacement2.cpp -- follow-up to abplacement: which parts of a big allocation can be written concurrently at more than the
// single-stream rate?  Two concurrent 4-GiB write streams (fl_fill_random on two hipStreams): region 0 paired with every
// other 4-GiB region of a 200-GiB allocation, then a few other anchors.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <hip/hip_runtime_api.h>
#include "fastlanes_amd.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv)
{
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
