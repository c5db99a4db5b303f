// Scratch (GPU box): on SLOW and FAST output allocations alike, how does the u32 W=7 unpack react to the XCD map's chunk size
// K (XCD x owns runs of K consecutive 32-block tiles; K = all -> one contiguous eighth per XCD, the shipped map; K = 1 ->
// plain round robin)?  If a slow allocation is slow because the 8 lock-step streams collide in the channel hash, changing
// their spacing (K) should move it.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "fl_device.hpp"
using namespace fl;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
struct Args { const u32x4* in; u32x4* out; uint64_t n_blocks; };

template <typename T, int W, int MAXW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, MAXW))) void k_unpack_x(Args a, uint64_t K)
{
    constexpr int BPW = 32;
    const uint64_t n_wg = (a.n_blocks + BPW - 1) / BPW;
    const uint64_t b = blockIdx.x;
    const uint64_t span = 8ull * K;
    const uint64_t wg = (b / span) * span + (b % 8) * K + (b / 8) % K;
    if (wg >= n_wg) return;
    const unsigned tid = threadIdx.x;
    const uint64_t blk = wg * BPW + (tid >> 3);
    const unsigned c = tid & 7u;
    if (blk >= a.n_blocks) return;
    Cell<T> in[W];
    const u32x4* pk = a.in + blk * (uint64_t)(8 * W) + c;
    static_for<W>([&](auto Wd) { in[decltype(Wd)::value] = load_cell<T, false>(pk + 8 * decltype(Wd)::value); });
    u32x4* wg_out = a.out + wg * (uint64_t)(BPW * Elem<T>::CELLS_PER_BLOCK);
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)wg_out, 0, BPW * Elem<T>::CELLS_PER_BLOCK * 16, 0x00020000);
    const unsigned vo = (tid >> 3) * (Elem<T>::CELLS_PER_BLOCK * 16) + c * 16;
    unpack_rows_by_address<T, W>(in, [&](auto R, const Cell<T>& v) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, vo + 16 * Elem<T>::row_cell(decltype(R)::value), 0, 18);
    });
}
__global__ void k_fill(uint64_t* p, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; p[i] = z ^ (z >> 31);
    }
}
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const uint64_t n = 10000000ull, n_wg = (n + 31) / 32;
    u32x4* in; CK(hipMalloc(&in, n * 896));
    hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, 0, (uint64_t*)in, n * 896 / 8);
    std::vector<u32x4*> outs(5);
    for (auto& o : outs) CK(hipMalloc(&o, n * 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint64_t G = (n_wg + 7) / 8;
    const uint64_t Ks[] = {1, 16, 256, 4096, 39063 /* ~G/1 -> 1/8 */ , G / 4, G / 2, G};
    printf("u32 W=7 unpack (cell-column, plain shape), GB/s; columns: XCD-map chunk K =");
    for (uint64_t K : Ks) printf(" %llu", (unsigned long long)K);
    printf("  (last = one contiguous eighth per XCD)\n");
    for (size_t k = 0; k < outs.size(); ++k) {
        printf("out%zu:", k);
        for (uint64_t K : Ks) {
            Args a{in, outs[k], n};
            const uint64_t span = 8 * K, grid = (n_wg + span - 1) / span * span;
            std::vector<float> ms;
            for (int r = 0; r < 6; ++r) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL((k_unpack_x<uint32_t, 7, 2>), dim3((unsigned)grid), dim3(256), 0, 0, a, K);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            printf(" %6.0f", n * 4992.0 / ms[2] / 1e6);
        }
        printf("\n");
    }
    return 0;
}
