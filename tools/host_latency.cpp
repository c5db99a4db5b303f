// host_latency.cpp -- what one trait-method call costs through the HOST tier of the C ABI (no Python in the loop),
// and what a large host-slice call sustains.  Run on the GPU box.
//   g++ -O2 -std=c++17 -I include -I /opt/rocm/include tools/host_latency.cpp -L fastlanes_amd -lfastlanes_amd -L /opt/rocm/lib -lamdhip64 \
//       -Wl,-rpath,'$ORIGIN/../fastlanes_amd' -Wl,-rpath,/opt/rocm/lib -o tools/host_latency
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include "fastlanes_amd.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <typename F> static double per_call_us(int reps, F&& f)
{
    for (int i = 0; i < 50; ++i) f();
    const double t0 = now();
    for (int i = 0; i < reps; ++i) f();
    return (now() - t0) / reps * 1e6;
}

int main()
{
    std::vector<uint16_t> v16(1024), p16(192), u16(1024);
    for (int i = 0; i < 1024; ++i) v16[i] = i % 8;
    if (fl_u16_pack_host(3, v16.data(), p16.data(), 1) != 0) { std::printf("no GPU / error %d\n", fl_last_hip_error()); return 2; }
    std::printf("single-block pack   u16 W=3  (benches/bitpacking.rs:19 shape): %6.1f us per call\n",
                per_call_us(5000, [&] { fl_u16_pack_host(3, v16.data(), p16.data(), 1); }));
    std::printf("single-block unpack u16 W=3  (benches/bitpacking.rs:43 shape): %6.1f us per call\n",
                per_call_us(5000, [&] { fl_u16_unpack_host(3, p16.data(), u16.data(), 1); }));
    std::printf("  round trip %s\n", std::memcmp(v16.data(), u16.data(), 2048) == 0 ? "ok" : "WRONG");
    std::vector<uint32_t> p32(224, 0x9E3779B9u), u32(1024);
    std::printf("single-block unpack u32 W=7:                                   %6.1f us per call\n",
                per_call_us(5000, [&] { fl_u32_unpack_host(7, p32.data(), u32.data(), 1); }));
    uint16_t one = 0;
    std::printf("unpack_single       u16 W=3  (benches/bitpacking.rs:57 shape): %6.1f us per call\n",
                per_call_us(5000, [&] { fl_u16_unpack_single_host(3, p16.data(), 1, 777, &one); }));
    for (size_t n : {64ul, 4096ul, 65536ul, 524288ul}) {
        std::vector<uint32_t> pk(n * 224), un(n * 1024);
        for (size_t i = 0; i < pk.size(); ++i) pk[i] = (uint32_t)(i * 2654435761u);
        std::memset(un.data(), 1, un.size() * 4);                     // touch the pages
        fl_u32_unpack_host(7, pk.data(), un.data(), n);
        const int reps = n >= 65536 ? 5 : 50;
        double best = 1e9;
        for (int r = 0; r < reps; ++r) {
            const double t0 = now();
            fl_u32_unpack_host(7, pk.data(), un.data(), n);
            const double dt = now() - t0;
            if (dt < best) best = dt;
        }
        std::printf("unpack u32 W=7 host slices, %7zu blocks (%8.1f MB in + out): %9.1f us  %6.2f GB/s  %6.2f Gint/s\n", n,
                    n * 4992 / 1e6, best * 1e6, n * 4992 / best / 1e9, n * 1024 / best / 1e9);
    }
    fl_host_release();
    // DEVICE tier on small columns (launch-bound): data resident in HBM, one call per column
    {
        uint32_t *d_in = nullptr, *d_out = nullptr;
        hipStream_t st;
        if (hipMalloc((void**)&d_in, 4096 * 896) != hipSuccess || hipMalloc((void**)&d_out, 4096 * 4096) != hipSuccess ||
            hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { std::printf("hipMalloc failed\n"); return 2; }
        (void)hipMemset(d_in, 0x5a, 4096 * 896);
        for (size_t n : {1ul, 64ul, 1024ul, 4096ul}) {
            const double sync_us = per_call_us(3000, [&] { fl_u32_unpack(7, d_in, d_out, n, st); (void)hipStreamSynchronize(st); });
            for (int i = 0; i < 50; ++i) fl_u32_unpack(7, d_in, d_out, n, st);
            (void)hipStreamSynchronize(st);
            const double t0 = now();
            for (int i = 0; i < 5000; ++i) fl_u32_unpack(7, d_in, d_out, n, st);
            (void)hipStreamSynchronize(st);
            const double async_us = (now() - t0) / 5000 * 1e6;
            std::printf("device tier unpack u32 W=7, %5zu blocks: launch + sync %6.1f us;  back-to-back async %6.2f us per call  (%7.1f Gint/s)\n",
                        n, sync_us, async_us, n * 1024 / async_us / 1e3);
        }
        (void)hipFree(d_in); (void)hipFree(d_out); (void)hipStreamDestroy(st);
    }
    return 0;
}
