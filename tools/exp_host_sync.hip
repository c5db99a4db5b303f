// exp_host_sync.hip -- what is the cheapest way for the host tier to learn that its one-block kernel has finished?
// The host tier's per-call cost (13 us) is the runtime's launch + hipStreamSynchronize (12 us for an empty device-tier call,
// profiles/r04_host_latency.txt).  Variants, same zero-copy kernel (reads 1 KiB of pinned host memory, writes 4 KiB of it):
//   A  launch + hipStreamSynchronize                                  (what fl_capi.hip's host_run does)
//   B  launch + hipEventRecord + hipEventSynchronize
//   C  launch + spin on hipStreamQuery
//   D  launch + a second one-thread kernel that stores a sequence number to pinned memory + host spins on it
//   E  the kernel itself stores the sequence number (system-scope release) as its last act + host spins on it
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp_host_sync.hip -o tools/exp_host_sync
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %d (%s) line %d\n", (int)e_, hipGetErrorString(e_), __LINE__); return 2; } } while (0)

__global__ void k_work(const uint32_t* in, uint32_t* out, volatile uint64_t* flag, uint64_t seq)
{
    const unsigned t = threadIdx.x;                       // 256 threads: 1 KiB in, 4 KiB out
    const uint32_t v = in[t];
    for (int k = 0; k < 4; ++k) out[k * 256 + t] = v + k;
    if (flag) {
        __syncthreads();
        if (t == 0) {
            __threadfence_system();
            __hip_atomic_store(const_cast<uint64_t*>(flag), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ void k_signal(uint64_t* flag, uint64_t seq)
{
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    char* pin = nullptr;
    CK(hipHostMalloc((void**)&pin, 65536, hipHostMallocDefault));
    uint32_t* in = (uint32_t*)pin;
    uint32_t* out = (uint32_t*)(pin + 4096);
    uint64_t* flag = (uint64_t*)(pin + 32768);
    *flag = 0;
    for (int i = 0; i < 256; ++i) in[i] = i;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    uint64_t seq = 0;
    const int REPS = 20000;
    auto spin = [&](uint64_t want) {
        for (uint64_t it = 0; __atomic_load_n(flag, __ATOMIC_ACQUIRE) != want; ++it)
            if ((it & 0xfffff) == 0xfffff && now() > 0 && it > (1ull << 32)) { std::printf("flag never arrived\n"); std::exit(3); }   // ~ seconds: never hang the box
    };
    for (int variant = 0; variant < 5; ++variant) {
        double t0 = 0;
        for (int i = -200; i < REPS; ++i) {
            if (i == 0) t0 = now();
            in[0] = (uint32_t)i;
            ++seq;
            switch (variant) {
            case 0: hipLaunchKernelGGL(k_work, dim3(1), dim3(256), 0, s, in, out, nullptr, 0); CK(hipStreamSynchronize(s)); break;
            case 1: hipLaunchKernelGGL(k_work, dim3(1), dim3(256), 0, s, in, out, nullptr, 0); CK(hipEventRecord(ev, s)); CK(hipEventSynchronize(ev)); break;
            case 2: hipLaunchKernelGGL(k_work, dim3(1), dim3(256), 0, s, in, out, nullptr, 0); while (hipStreamQuery(s) == hipErrorNotReady) { } break;
            case 3: hipLaunchKernelGGL(k_work, dim3(1), dim3(256), 0, s, in, out, nullptr, 0); hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, s, flag, seq); spin(seq); break;
            default: hipLaunchKernelGGL(k_work, dim3(1), dim3(256), 0, s, in, out, flag, seq); spin(seq); break;
            }
            if (out[0] != (uint32_t)i || out[3 * 256 + 255] != 255u + 3u) { std::printf("variant %d: stale output at call %d\n", variant, i); return 1; }
        }
        const double us = (now() - t0) / REPS * 1e6;
        static const char* names[] = {"A launch + hipStreamSynchronize", "B launch + event record + hipEventSynchronize", "C launch + spin on hipStreamQuery",
                                      "D launch + signal kernel + spin on a pinned flag", "E kernel stores the flag itself + spin"};
        std::printf("%-52s %6.2f us per call\n", names[variant], us);
    }
    CK(hipStreamSynchronize(s));
    return 0;
}
