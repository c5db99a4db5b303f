// Scratch (GPU box): does WHICH XCD writes WHICH eighth of a buffer matter?  A bare sc1|nt write stream over a 40 GB buffer
// with the XCD-contiguous map, the eighths assigned to the XCDs through a permutation.  If the physical placement effect is
// an XCD <-> memory-region affinity, some permutations are fast and some slow on the same buffer.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct Perm { unsigned char p[8]; };

template <int MAXW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, MAXW)))
void k_write(u32x4* out, uint64_t n_tiles, uint64_t tiles_per_xcd, Perm perm, int linear)
{
    uint64_t tile;
    if (linear) tile = blockIdx.x;
    else tile = (uint64_t)perm.p[blockIdx.x & 7u] * tiles_per_xcd + (blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(out + tile * (uint64_t)(256 * 32)), 0, 256 * 32 * 16, 0x00020000);
    u32x4 v = {(uint32_t)tile * 2654435761u, threadIdx.x * 40503u, (uint32_t)(tile >> 3) ^ 0x9E3779B9u, 0x85EBCA6Bu * threadIdx.x};
#pragma unroll
    for (int i = 0; i < 32; ++i) { v.x += 0x61C88647u; __builtin_amdgcn_raw_buffer_store_b128(v, rs, (i * 256 + threadIdx.x) * 16, 0, 18); }
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const uint64_t bytes = 40ull << 30, n_tiles = bytes / (256 * 32 * 16), tpx = (n_tiles + 7) / 8;
    std::vector<u32x4*> bufs(5);
    for (auto& b : bufs) CK(hipMalloc(&b, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](u32x4* b, Perm p, int linear) {
        std::vector<float> ms;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_write<3>, dim3(linear ? (unsigned)n_tiles : (unsigned)(tpx * 8)), dim3(256), 0, 0, b, n_tiles, tpx, p, linear);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        return bytes / ms[2] / 1e6;
    };
    Perm id{{0, 1, 2, 3, 4, 5, 6, 7}};
    printf("write-only stream, 40 GiB buffers, GB/s: identity XCD map | rotations 1..7 | reversed | two random | linear (no XCD map)\n");
    for (size_t k = 0; k < bufs.size(); ++k) {
        printf("buf%zu: id %6.0f | rot", k, run(bufs[k], id, 0));
        for (int s = 1; s < 8; ++s) { Perm p; for (int x = 0; x < 8; ++x) p.p[x] = (x + s) & 7; printf(" %6.0f", run(bufs[k], p, 0)); }
        Perm rev{{7, 6, 5, 4, 3, 2, 1, 0}}, r1{{3, 6, 0, 5, 2, 7, 4, 1}}, r2{{5, 2, 7, 0, 3, 6, 1, 4}};
        printf(" | rev %6.0f | rnd %6.0f %6.0f | linear %6.0f\n", run(bufs[k], rev, 0), run(bufs[k], r1, 0), run(bufs[k], r2, 0), run(bufs[k], id, 1));
    }
    return 0;
}
