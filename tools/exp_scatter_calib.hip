// exp_scatter_calib.hip -- what does a SCATTERED read cost the memory, and what does FETCH_SIZE say about it?  (the x2.0
// read calibration of rocprofv3's FETCH_SIZE on gfx950 comes from a coalesced copy; batched unpack_single's random lookups were quoted with it.)
// N random 128-byte-aligned lines of a 16-GiB buffer (far beyond the 256-MiB Infinity Cache); per access a lane group reads
//   4 B (one lane), 16 B (one lane), 32 B (2 lanes), 64 B (4 lanes) or the whole 128-B line (8 lanes x 16 B)
// of its line and folds it into a checksum.  If the memory moves whole lines whatever is asked for, every variant takes the same time per
// access; if it moves 32- or 64-byte sectors, the narrow ones are faster.  Under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` the raw counter
// per access next to the bytes the full-line variant certainly moved (128) gives the scattered calibration.
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp_scatter_calib tools/exp_scatter_calib.hip ; tools/exp_scatter_calib [accesses=2^28]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

// LANES lanes share one access (one random line); each reads BYTES_PER_LANE bytes at lane_in_group * BYTES_PER_LANE
template <int LANES, int BYTES_PER_LANE>
__global__ __launch_bounds__(256) void k_scatter(const char* buf, uint64_t n_lines, uint64_t n_access, uint64_t seed, uint64_t* sink)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t access = tid / LANES;
    const unsigned sub = (unsigned)(tid % LANES);
    uint64_t acc = 0;
    if (access < n_access) {
        const uint64_t line = mix(access * 0x9e3779b97f4a7c15ull + seed) % n_lines;
        const char* p = buf + line * 128 + sub * BYTES_PER_LANE;
        if constexpr (BYTES_PER_LANE == 4) acc = *reinterpret_cast<const uint32_t*>(p);
        else {
            const uint4 v = *reinterpret_cast<const uint4*>(p);
            acc = v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if ((uint32_t)acc == 0x12345678u) sink[0] = acc;         // never true (the buffer holds 0x01 bytes): keeps the loads alive
}

// the coalesced reference: every lane reads 16 consecutive bytes, the whole grid a contiguous range
__global__ __launch_bounds__(256) void k_stream(const char* buf, uint64_t n_bytes, uint64_t* sink)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t acc = 0;
    if (tid * 16 < n_bytes) {
        const uint4 v = *reinterpret_cast<const uint4*>(buf + tid * 16);
        acc = v.x ^ v.y ^ v.z ^ v.w;
    }
    if ((uint32_t)acc == 0x12345678u) sink[0] = acc;
}

template <class F> static float median_ms(F&& f, int reps = 5)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> ms;
    for (int i = -1; i < reps; ++i) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float t; CK(hipEventElapsedTime(&t, a, b));
        if (i >= 0) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main(int argc, char** argv)
{
    const uint64_t n_access = argc > 1 ? strtoull(argv[1], nullptr, 0) : (1ull << 28);
    const uint64_t bytes = 16ull << 30, n_lines = bytes / 128;
    char* buf; uint64_t* sink;
    CK(hipMalloc((void**)&buf, bytes)); CK(hipMalloc((void**)&sink, 8));
    CK(hipMemset(buf, 1, bytes)); CK(hipMemset(sink, 0, 8));
    printf("%llu scattered accesses to random 128-B lines of a 16-GiB buffer; ms, G accesses/s, GB/s if each access moved a whole 128-B line\n", (unsigned long long)n_access);
    auto run = [&](const char* name, auto kernel, int lanes) {
        const uint64_t threads = n_access * lanes;
        const unsigned grid = (unsigned)((threads + 255) / 256);
        uint64_t seed = 1;
        float ms = median_ms([&] { hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, buf, n_lines, n_access, seed++, sink); });
        printf("%-44s %8.3f ms  %7.2f G accesses/s  %7.0f GB/s at 128 B per access\n", name, ms, n_access / ms / 1e6, n_access * 128.0 / ms / 1e6);
    };
    run("k_scatter<1,4>   4 B of the line (1 lane)", k_scatter<1, 4>, 1);
    run("k_scatter<1,16> 16 B of the line (1 lane)", k_scatter<1, 16>, 1);
    run("k_scatter<2,16> 32 B of the line (2 lanes)", k_scatter<2, 16>, 2);
    run("k_scatter<4,16> 64 B of the line (4 lanes)", k_scatter<4, 16>, 4);
    run("k_scatter<8,16> the whole 128-B line (8 lanes)", k_scatter<8, 16>, 8);
    {
        const uint64_t nb = 8ull << 30;
        const unsigned grid = (unsigned)(nb / 16 / 256);
        float ms = median_ms([&] { hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, buf, nb, sink); });
        printf("%-44s %8.3f ms  %7.0f GB/s (8 GiB, coalesced: the x2.0 calibration's pattern)\n", "k_stream", ms, nb / ms / 1e6);
    }
    return 0;
}
