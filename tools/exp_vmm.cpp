// exp_vmm.cpp -- round 6: CONSTRUCT a placement instead of probing for one (VERDICT r05 "next" #1).
//
// The HIP virtual-memory API (hipMemCreate / hipMemAddressReserve / hipMemMap / hipMemSetAccess) separates physical chunks from
// the addresses they appear at.  This tool
//   1. creates a pool of physical chunks (default 144 x 1 GiB) and maps them back to back,
//   2. classifies every chunk by the thin-write interference test of profiles/exp_region_map_r03.txt (a bulk read of a representative
//      chunk next to a thin write stream into chunk g is slow iff g is of the representative's class of memory),
//   3. re-maps chunks into (input, output) address ranges under a list of LAYOUTS -- which class feeds the input, which classes the
//      output cycles through, at what granularity -- and times the codec kernel and a bare stream of its bytes on each, round-robin.
// Usage: exp_vmm [pool_GiB=144] [chunk_MiB=1024] [workload=unpack32w7|pack32w7|mixed32] [n_blocks=10000000] [rounds=2] [va_align_MiB=2] [slab_GiB=0]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <hip/hip_runtime_api.h>
#include "fastlanes_amd.h"
#include "fastlanes_amd_internal.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s (%d) at line %d: %s\n", hipGetErrorString(e_), (int)e_, __LINE__, #x); fflush(stdout); exit(1); } } while (0)
#define FL(x) do { int r_ = (x); if (r_ != 0) { printf("fastlanes error %d (%s) at line %d: %s\n", r_, fl_status_string(r_), __LINE__, #x); fflush(stdout); exit(1); } } while (0)

static const size_t MiB = 1ull << 20, GiB = 1ull << 30;
static hipEvent_t E0, E1;
static int DEV = 0;

struct Pool {
    size_t chunk = 0;
    std::vector<hipMemGenericAllocationHandle_t> h;
    std::vector<int> cls;
};

static void map_chunks(char* va, const Pool& p, const std::vector<int>& idx)
{
    hipMemAccessDesc acc{};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = DEV;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (size_t i = 0; i < idx.size(); ++i) CK(hipMemMap(va + i * p.chunk, p.chunk, 0, p.h[idx[i]], 0));
    if (!idx.empty()) CK(hipMemSetAccess(va, idx.size() * p.chunk, &acc, 1));
}
static void unmap_chunks(char* va, const Pool& p, size_t n)
{
    for (size_t i = 0; i < n; ++i) CK(hipMemUnmap(va + i * p.chunk, p.chunk));
}

template <class F>
static float median_ms(int reps, F&& launch)
{
    std::vector<float> ms;
    for (int i = -1; i < reps; ++i) {
        CK(hipEventRecord(E0, nullptr));
        launch();
        CK(hipEventRecord(E1, nullptr));
        CK(hipEventSynchronize(E1));
        float t;
        CK(hipEventElapsedTime(&t, E0, E1));
        if (i >= 0) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

// classes of the n pieces of `chunk` bytes at `base` by the thin-write interference test: unpack_compare u32 W=20 reads `chunk` bytes
// of a representative piece, its 1/20 mask goes into the middle of piece g; slow = g is of the representative's class
static int classify(char* base, size_t n, size_t chunk, std::vector<int>& cls)
{
    fl_internal_set_kernel_policy(33554432 * 31);                       // whole-column tile map, as the class map was characterised
    const unsigned PW = 20;
    const size_t probe_blocks = std::min(chunk, (size_t)4 << 30) / (128 * PW);
    auto probe = [&](size_t rep, size_t g) {
        const uint32_t* src = (const uint32_t*)(base + rep * chunk);
        uint32_t* mask = (uint32_t*)(base + g * chunk + chunk / 2);
        return median_ms(5, [&] { FL(fl_u32_unpack_compare(PW, src, FL_CMP_LT, 1u << (PW - 1), probe_blocks, mask, nullptr)); });
    };
    cls.assign(n, -1);
    std::vector<float> ms(n);
    float threshold = 0.f;
    int n_classes = 0;
    for (int c = 0; c < 4; ++c) {
        size_t rep = 0;
        while (rep < n && cls[rep] != -1) ++rep;
        if (rep == n) break;
        cls[rep] = c;
        n_classes = c + 1;
        float lo = 1e30f, hi = 0.f;
        for (size_t g = 0; g < n; ++g) {
            if (cls[g] != -1) continue;
            ms[g] = probe(rep, g);
            lo = std::min(lo, ms[g]);
            hi = std::max(hi, ms[g]);
        }
        if (hi == 0.f) break;
        printf("class %c: representative piece %zu, probe %.3f .. %.3f ms (%.0f .. %.0f GB/s)\n", 'A' + c, rep, lo, hi,
               probe_blocks * 128.0 * (PW + 1) / hi / 1e6, probe_blocks * 128.0 * (PW + 1) / lo / 1e6);
        if (threshold == 0.f) {
            if (hi - lo <= 0.05f * hi) { printf("  one level only: no class structure visible at this piece size\n"); break; }
            threshold = 0.5f * (lo + hi);
        }
        int hist[10] = {0};                                  // is the map binary at this piece size?
        for (size_t g = 0; g < n; ++g)
            if (cls[g] == -1) hist[std::min(9, (int)((ms[g] - lo) / (hi - lo + 1e-9f) * 10))]++;
        printf("  histogram lo..hi:");
        for (int b : hist) printf(" %d", b);
        printf("\n");
        for (size_t g = 0; g < n; ++g)
            if (cls[g] == -1 && ms[g] > threshold) cls[g] = c;
    }
    fl_internal_set_kernel_policy(0);
    return n_classes;
}
static void print_classes(const std::vector<int>& cls)
{
    printf("  ");
    for (size_t g = 0; g < cls.size(); ++g) {
        putchar(cls[g] < 0 ? '?' : 'A' + cls[g]);
        if (g % 64 == 63) printf("\n  ");
    }
    printf("\n");
}


// ---- mode "position": does WHERE in the device memory (and in what chunk size / order) matter, classes aside? -------------------------
// For every chunk size: a pool over most of the device in creation order, its class map per GiB, unpack u32 W=7 (input 9 GiB, output right
// behind it) at every 24-GiB offset of the pool, then the first 48 GiB of the pool re-mapped in reverse and in shuffled chunk order.
static void position_mode(size_t pool_gib, const std::vector<size_t>& chunk_mibs, size_t n_blocks, const hipMemAllocationProp& prop)
{
    const size_t ib = n_blocks * 128 * 7, ob = n_blocks * 4096;
    const size_t in_gib = (ib + GiB - 1) / GiB, out_gib = (ob + GiB - 1) / GiB, win_gib = in_gib + out_gib;
    auto timed = [&](const char* in, char* out) {
        float k = median_ms(7, [&] { FL(fl_u32_unpack(7, (const uint32_t*)in, (uint32_t*)out, n_blocks, nullptr)); });
        return (ib + ob) / k / 1e6;
    };
    if (!getenv("EXP_NO_PAIR")) {
        void *a = nullptr, *b = nullptr;
        CK(hipMalloc(&a, ib));
        CK(hipMalloc(&b, ob));
        FL(fl_fill_random(a, ib, 11, nullptr));
        const double g = timed((const char*)a, (char*)b);
        printf("two hipMallocs on the empty device: %5.0f GB/s (%.3f)\n", g, g / 8000);
        CK(hipFree(a));
        CK(hipFree(b));
    }
    for (size_t cm : chunk_mibs) {
        Pool p;
        p.chunk = cm * MiB;
        const size_t n = pool_gib * GiB / p.chunk, per_gib = GiB / p.chunk ? GiB / p.chunk : 1;
        p.h.resize(n);
        for (size_t i = 0; i < n; ++i) CK(hipMemCreate(&p.h[i], p.chunk, &prop, 0));
        char* va = nullptr;
        CK(hipMemAddressReserve((void**)&va, n * p.chunk, 2 * MiB, nullptr, 0));
        std::vector<int> ident(n);
        for (size_t i = 0; i < n; ++i) ident[i] = (int)i;
        map_chunks(va, p, ident);
        FL(fl_fill_random(va, n * p.chunk, 3, nullptr));
        CK(hipDeviceSynchronize());
        printf("== chunk %zu MiB: %zu chunks mapped in creation order at %p; class of every GiB of the pool:\n", cm, n, (void*)va);
        std::vector<int> cls;
        classify(va, pool_gib, GiB, cls);
        print_classes(cls);
        printf("unpack u32 W=7, %zu blocks: input at <GiB> of the pool, output right behind it: GB/s (of 8 TB/s) [class of every 2nd GiB: in | out]\n", n_blocks);
        for (size_t o = 0; o + win_gib <= pool_gib; o += 24) {
            FL(fl_fill_random(va + o * GiB, ib, 11, nullptr));
            const double g = timed(va + o * GiB, va + (o + in_gib) * GiB);
            printf("  @%3zu: %5.0f (%.3f)  [", o, g, g / 8000);
            for (size_t q = o; q < o + in_gib; q += 2) putchar(cls[q] < 0 ? '?' : 'A' + cls[q]);
            printf(" | ");
            for (size_t q = o + in_gib; q < o + win_gib; q += 2) putchar(cls[q] < 0 ? '?' : 'A' + cls[q]);
            printf("]\n");
        }
        fflush(stdout);
        unmap_chunks(va, p, n);
        // the first window's chunks in other orders (same physical memory, other addresses)
        const size_t nin = (in_gib * GiB + p.chunk - 1) / p.chunk, nw = nin + (out_gib * GiB + p.chunk - 1) / p.chunk;
        char* vw = nullptr;
        CK(hipMemAddressReserve((void**)&vw, nw * p.chunk, 2 * MiB, nullptr, 0));
        for (int variant = 0; variant < 4 && nw > 1; ++variant) {
            std::vector<int> idx(nw);
            for (size_t i = 0; i < nw; ++i) idx[i] = (int)i;
            const char* name = "creation order again";
            if (variant == 1) { std::reverse(idx.begin(), idx.end()); name = "reverse order (input = the last chunks created)"; }
            if (variant == 2) {
                uint64_t x = 88172645463325252ull;
                for (size_t i = nw - 1; i > 0; --i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; std::swap(idx[i], idx[x % (i + 1)]); }
                name = "shuffled";
            }
            if (variant == 3) {                                 // input as created, output chunks interleaved from its two halves
                const size_t no = nw - nin, h = no / 2;
                for (size_t i = 0; i < no; ++i) idx[nin + i] = (int)(nin + (i % 2 ? h + i / 2 : i / 2));
                name = "output = its two halves interleaved chunk by chunk";
            }
            map_chunks(vw, p, idx);
            FL(fl_fill_random(vw, ib, 11, nullptr));
            const double g = timed(vw, vw + nin * p.chunk);
            printf("  first window, %-52s %5.0f (%.3f)\n", name, g, g / 8000);
            CK(hipDeviceSynchronize());
            unmap_chunks(vw, p, nw);
        }
        // chunks CHOSEN by class from the whole pool (1-GiB chunks only: the class map above is per GiB), next to creation order, twice
        if (p.chunk == GiB) {
            std::vector<std::vector<int>> by(4);
            for (size_t g = 0; g < n; ++g)
                if (cls[g] >= 0) by[cls[g]].push_back((int)g);
            struct L { const char* name; const char* in; const char* out; };
            const L ls[] = {{"in A | out A", "A", "A"}, {"in A | out B", "A", "B"}, {"in A | out BC alternating", "A", "BC"}, {"in A | out ABC", "A", "ABC"},
                            {"in ABC | out ABC", "ABC", "ABC"}, {"creation order", "", ""}};
            for (int pass = 0; pass < 2; ++pass)
                for (const L& l : ls) {
                    std::vector<int> idx;
                    size_t next[4] = {0, 0, 0, 0};
                    bool ok = true;
                    if (!l.in[0]) { for (size_t i = 0; i < nw; ++i) idx.push_back((int)i); }
                    else {
                        for (size_t i = 0; i < nw && ok; ++i) {
                            const char* pat = i < nin ? l.in : l.out;
                            const int c = pat[(i < nin ? i : i - nin) % strlen(pat)] - 'A';
                            if (next[c] >= by[c].size()) ok = false; else idx.push_back(by[c][next[c]++]);
                        }
                    }
                    if (!ok) { printf("  chosen by class, %-28s (not enough chunks)\n", l.name); continue; }
                    map_chunks(vw, p, idx);
                    FL(fl_fill_random(vw, ib, 11, nullptr));
                    const double g = timed(vw, vw + nin * p.chunk);
                    char* o = vw + nin * p.chunk;
                    const size_t nu = ob / 4096;
                    const float w31 = median_ms(5, [&] { FL(fl_internal_bare_stream(nullptr, 0, nullptr, 0, o, 4096, nu, 1, 6, 31, nullptr)); });
                    const float w18 = median_ms(5, [&] { FL(fl_internal_bare_stream(nullptr, 0, nullptr, 0, o, 4096, nu, 1, 6, 18, nullptr)); });
                    const float r31 = median_ms(5, [&] { FL(fl_internal_bare_stream(o, 4096, nullptr, 0, vw, 0, nu, 1, 6, 31, nullptr)); });
                    const float r18 = median_ms(5, [&] { FL(fl_internal_bare_stream(o, 4096, nullptr, 0, vw, 0, nu, 1, 6, 18, nullptr)); });
                    printf("  chosen by class, %-28s %5.0f (%.3f)  output buffer alone: write %5.0f (1-GiB windows %5.0f)  read %5.0f (windows %5.0f)\n", l.name, g, g / 8000,
                           ob / w31 / 1e6, ob / w18 / 1e6, ob / r31 / 1e6, ob / r18 / 1e6);
                    fflush(stdout);
                    CK(hipDeviceSynchronize());
                    unmap_chunks(vw, p, nw);
                }
        }
        CK(hipMemAddressFree(vw, nw * p.chunk));
        CK(hipMemAddressFree(va, n * p.chunk));
        for (size_t i = 0; i < n; ++i) CK(hipMemRelease(p.h[i]));
        fflush(stdout);
    }
}



// ---- mode "state": is the 0.78 / 0.85 split a STATE of the device rather than a place in memory? ----------------------------------------
// (the va runs: the first measurement of a process 0.78, the same addresses two measurements later 0.85)
#include <chrono>
#include <thread>
#include <fstream>
#include <glob.h>
static std::string active_clocks()
{
    std::string out;
    for (const char* f : {"pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"}) {
        glob_t g;
        std::string pat = std::string("/sys/class/drm/card*/device/") + f;
        if (glob(pat.c_str(), 0, nullptr, &g) == 0) {
            for (size_t i = 0; i < g.gl_pathc && i < 1; ++i) {
                std::ifstream in(g.gl_pathv[i]);
                std::string l;
                while (std::getline(in, l))
                    if (l.find('*') != std::string::npos) out += std::string(f + 7) + "=" + l + " ";
            }
        }
        globfree(&g);
    }
    glob_t g;
    if (glob("/sys/class/drm/card*/device/hwmon/hwmon*/temp*_input", 0, nullptr, &g) == 0) {
        for (size_t i = 0; i < g.gl_pathc; ++i) {
            std::string path = g.gl_pathv[i], lab = path.substr(0, path.size() - 5) + "label", l, v;
            std::ifstream a(lab), b(path);
            std::getline(a, l);
            std::getline(b, v);
            if (!v.empty()) out += (l.empty() ? "t" : l) + "=" + std::to_string(atol(v.c_str()) / 1000) + "C ";
        }
    }
    globfree(&g);
    if (glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_*", 0, nullptr, &g) == 0) {
        for (size_t i = 0; i < g.gl_pathc; ++i) {
            std::string path = g.gl_pathv[i], v;
            if (path.find("average") == std::string::npos && path.find("input") == std::string::npos) continue;
            std::ifstream b(path);
            std::getline(b, v);
            if (!v.empty()) out += "P=" + std::to_string(atol(v.c_str()) / 1000000) + "W ";
        }
    }
    globfree(&g);
    return out.empty() ? "(no readable pp_dpm_* files)" : out;
}
static void state_mode(size_t n_blocks)
{
    const size_t ib = n_blocks * 128 * 7, ob = n_blocks * 4096;
    void *a = nullptr, *b = nullptr;
    CK(hipMalloc(&a, ib));
    CK(hipMalloc(&b, ob));
    FL(fl_fill_random(a, ib, 11, nullptr));
    CK(hipDeviceSynchronize());
    auto burst = [&](const char* what, int n) {
        printf("%-28s clocks before: %s\n   ms per launch:", what, active_clocks().c_str());
        std::vector<float> ms(n);
        std::vector<hipEvent_t> ev(n + 1);
        for (auto& e : ev) CK(hipEventCreate(&e));
        CK(hipEventRecord(ev[0], nullptr));
        for (int i = 0; i < n; ++i) {
            FL(fl_u32_unpack(7, (const uint32_t*)a, (uint32_t*)b, n_blocks, nullptr));
            CK(hipEventRecord(ev[i + 1], nullptr));
        }
        CK(hipEventSynchronize(ev[n]));
        const std::string during = active_clocks();
        for (int i = 0; i < n; ++i) { CK(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1])); printf(" %.2f", ms[i]); }
        for (auto& e : ev) CK(hipEventDestroy(e));
        printf("\n   clocks right after: %s\n", during.c_str());
        fflush(stdout);
    };
    burst("cold (first launches)", 40);
    for (double idle : {0.1, 1.0}) {
        std::this_thread::sleep_for(std::chrono::duration<double>(idle));
        char w[64];
        snprintf(w, sizeof w, "after %.2f s idle", idle);
        burst(w, 16);
    }
    // a FRESH pair of allocations in the warm state
    void *c = nullptr, *d = nullptr;
    CK(hipMalloc(&c, ib));
    CK(hipMalloc(&d, ob));
    FL(fl_fill_random(c, ib, 11, nullptr));
    std::swap(a, c);
    std::swap(b, d);
    burst("fresh second pair, warm", 24);
    CK(hipFree(c));
    CK(hipFree(d));
    burst("second pair, first freed", 16);
    void *e = nullptr, *f = nullptr;
    CK(hipMalloc(&e, ib));
    CK(hipMalloc(&f, ob));
    FL(fl_fill_random(e, ib, 11, nullptr));
    CK(hipFree(a));
    CK(hipFree(b));
    a = e;
    b = f;
    burst("third pair (reuses the first's memory?)", 24);
    CK(hipFree(a));
    CK(hipFree(b));
    // five pairs alive at once (walking down the device memory), each measured three times round-robin: is a pair's level its own?
    void *pi[5], *po[5];
    for (int k = 0; k < 5; ++k) { CK(hipMalloc(&pi[k], ib)); CK(hipMalloc(&po[k], ob)); FL(fl_fill_random(pi[k], ib, 11, nullptr)); }
    for (int r = 0; r < 2; ++r)
        for (int k = 0; k < 5; ++k) {
            a = pi[k];
            b = po[k];
            char w[64];
            snprintf(w, sizeof w, "pair %d of 5, pass %d", k, r);
            burst(w, 10);
        }
    // which side of a slow pair is slow?  write-only and read-only bare streams over each pair's OUTPUT buffer, and read-only over its input
    printf("per pair: unpack | write-only stream over the output buffer | read-only stream over it | 7:32 bare stream   (GB/s)\n");
    for (int k = 0; k < 5; ++k) {
        const size_t nu = ob / 4096;
        float tk = median_ms(5, [&] { FL(fl_u32_unpack(7, (const uint32_t*)pi[k], (uint32_t*)po[k], n_blocks, nullptr)); });
        float tw = median_ms(5, [&] { FL(fl_internal_bare_stream(nullptr, 0, nullptr, 0, po[k], 4096, nu, 1, 6, 31, nullptr)); });
        float tr = median_ms(5, [&] { FL(fl_internal_bare_stream(po[k], 4096, nullptr, 0, pi[k], 0, nu, 1, 6, 31, nullptr)); });
        float tm = median_ms(5, [&] { FL(fl_internal_bare_stream(pi[k], 896, nullptr, 0, po[k], 4096, nu, 1, 3, 31, nullptr)); });
        printf("  pair %d: %5.0f | %5.0f | %5.0f | %5.0f\n", k, (ib + ob) / tk / 1e6, ob / tw / 1e6, ob / tr / 1e6, (ib + ob) / tm / 1e6);
    }
    // is a buffer's write rate about the SPACING of the eight XCDs' write positions (length / 8 under the whole-column tile map, 2^w units / 8
    // under a windowed one) or about the memory itself?
    printf("write-only stream over the first L bytes of each pair's output buffer, whole-column tile map, L = full - {0, 0.25, 1, 2.5, 6, 19} GiB; then windows of 2^{12,16,18,20,22} units of 4 KiB\n");
    for (int k = 0; k < 5; ++k) {
        printf("  pair %d:", k);
        for (double cut : {0.0, 0.25, 1.0, 2.5, 6.0, 19.0}) {
            const size_t nu = (ob - (size_t)(cut * GiB)) / 4096;
            float tw = median_ms(5, [&] { FL(fl_internal_bare_stream(nullptr, 0, nullptr, 0, po[k], 4096, nu, 1, 6, 31, nullptr)); });
            printf(" %5.0f", nu * 4096.0 / tw / 1e6);
        }
        printf(" |");
        for (int w : {12, 16, 18, 20, 22}) {
            const size_t nu = ob / 4096;
            float tw = median_ms(5, [&] { FL(fl_internal_bare_stream(nullptr, 0, nullptr, 0, po[k], 4096, nu, 1, 6, w, nullptr)); });
            printf(" %5.0f", nu * 4096.0 / tw / 1e6);
        }
        printf("\n");
        fflush(stdout);
    }
    printf("write-only, whole-column map, the length cut by j units of 4 KiB (the spacing of the XCDs' positions changes by j/8 units): j = 0 8 24 64 200 1000 5000 30000 100000\n");
    for (int k = 0; k < 5; ++k) {
        printf("  pair %d:", k);
        for (size_t j : {0, 8, 24, 64, 200, 1000, 5000, 30000, 100000}) {
            const size_t nu = ob / 4096 - j;
            float tw = median_ms(5, [&] { FL(fl_internal_bare_stream(nullptr, 0, nullptr, 0, po[k], 4096, nu, 1, 6, 31, nullptr)); });
            printf(" %5.0f", nu * 4096.0 / tw / 1e6);
        }
        printf("\n");
        fflush(stdout);
    }
    printf("the same for READ-only streams over the output buffers\n");
    for (int k = 0; k < 5; ++k) {
        printf("  pair %d:", k);
        for (double cut : {0.0, 0.25, 1.0, 2.5, 6.0, 19.0}) {
            const size_t nu = (ob - (size_t)(cut * GiB)) / 4096;
            float tw = median_ms(5, [&] { FL(fl_internal_bare_stream(po[k], 4096, nullptr, 0, pi[k], 0, nu, 1, 6, 31, nullptr)); });
            printf(" %5.0f", nu * 4096.0 / tw / 1e6);
        }
        printf(" |");
        for (int w : {12, 16, 18, 20, 22}) {
            const size_t nu = ob / 4096;
            float tw = median_ms(5, [&] { FL(fl_internal_bare_stream(po[k], 4096, nullptr, 0, pi[k], 0, nu, 1, 6, w, nullptr)); });
            printf(" %5.0f", nu * 4096.0 / tw / 1e6);
        }
        printf("\n");
        fflush(stdout);
    }
}


// ---- mode "remap": does hipMemUnmap + hipMemMap of OTHER chunks at the same address really switch the physical memory behind it? -----
static void remap_mode(const hipMemAllocationProp& prop)
{
    Pool p;
    p.chunk = GiB;
    p.h.resize(4);
    for (auto& h : p.h) CK(hipMemCreate(&h, p.chunk, &prop, 0));
    char *V = nullptr, *V2 = nullptr;
    CK(hipMemAddressReserve((void**)&V, 2 * GiB, 2 * MiB, nullptr, 0));
    CK(hipMemAddressReserve((void**)&V2, 2 * GiB, 2 * MiB, nullptr, 0));
    auto first = [&](const char* at) { uint64_t w[2] = {0, 0}; CK(hipMemcpy(w, at, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(w + 1, at + GiB + 4096, 8, hipMemcpyDeviceToHost)); return std::pair<uint64_t, uint64_t>(w[0], w[1]); };
    map_chunks(V, p, {0, 1});
    FL(fl_fill_random(V, 2 * GiB, 1, nullptr));
    CK(hipDeviceSynchronize());
    auto w1 = first(V);
    unmap_chunks(V, p, 2);
    map_chunks(V, p, {2, 3});
    FL(fl_fill_random(V, 2 * GiB, 2, nullptr));
    CK(hipDeviceSynchronize());
    auto w2 = first(V);
    unmap_chunks(V, p, 2);
    map_chunks(V2, p, {0, 1});
    auto w3 = first(V2);
    map_chunks(V, p, {2, 3});
    auto w4 = first(V);
    printf("chunks {0,1} at V filled with seed 1: %016llx %016llx\n", (unsigned long long)w1.first, (unsigned long long)w1.second);
    printf("chunks {2,3} at V (after unmap + map) filled with seed 2: %016llx %016llx\n", (unsigned long long)w2.first, (unsigned long long)w2.second);
    printf("chunks {0,1} read back at V2: %016llx %016llx  -> %s\n", (unsigned long long)w3.first, (unsigned long long)w3.second,
           w3 == w1 ? "still seed 1: the second fill went to chunks {2,3}, the re-mapping is real" : w3 == w2 ? "SEED 2: the address kept pointing at chunks {0,1}" : "neither?!");
    printf("chunks {2,3} read back at V: %016llx %016llx -> %s\n", (unsigned long long)w4.first, (unsigned long long)w4.second, w4 == w2 ? "seed 2" : "NOT seed 2");
    // (b) the same with the address range FREED and reserved again in between (what an allocator that releases a pair does)
    unmap_chunks(V, p, 2);
    unmap_chunks(V2, p, 2);
    CK(hipMemAddressFree(V, 2 * GiB));
    CK(hipMemAddressFree(V2, 2 * GiB));
    char *W = nullptr, *W2 = nullptr;
    CK(hipMemAddressReserve((void**)&W, 2 * GiB, 2 * MiB, V, 0));           // ask for the same address back
    map_chunks(W, p, {0, 1});
    FL(fl_fill_random(W, 2 * GiB, 3, nullptr));
    CK(hipDeviceSynchronize());
    auto x1 = first(W);
    unmap_chunks(W, p, 2);
    CK(hipMemAddressFree(W, 2 * GiB));
    char* W1 = nullptr;
    CK(hipMemAddressReserve((void**)&W1, 2 * GiB, 2 * MiB, W, 0));
    map_chunks(W1, p, {2, 3});
    FL(fl_fill_random(W1, 2 * GiB, 4, nullptr));
    CK(hipDeviceSynchronize());
    auto x2 = first(W1);
    CK(hipMemAddressReserve((void**)&W2, 2 * GiB, 2 * MiB, nullptr, 0));
    map_chunks(W2, p, {0, 1});
    auto x3 = first(W2);
    printf("with hipMemAddressFree + hipMemAddressReserve in between: first range %p (was %p), again %p (%s)\n", (void*)W, (void*)V, (void*)W1, W1 == W ? "the SAME address" : "another address");
    printf("  chunks {0,1} filled with seed 3 %016llx, then chunks {2,3} at the re-reserved range filled with seed 4 %016llx; chunks {0,1} read back elsewhere: %016llx -> %s\n",
           (unsigned long long)x1.first, (unsigned long long)x2.first, (unsigned long long)x3.first,
           x3 == x1 ? "still seed 3: a re-RESERVED range maps what it is told to" : x3 == x2 ? "SEED 4: stale here too" : "neither?!");
}

// ---- mode "va": the SAME physical chunks at different ADDRESSES -------------------------------------------------------------------------
// (the position runs showed: the same top-of-memory bytes stream at 0.78 through two hipMallocs and at 0.86 through one mapped range)
static void va_mode(size_t n_blocks, size_t chunk_mib, const hipMemAllocationProp& prop)
{
    const size_t ib = n_blocks * 128 * 7, ob = n_blocks * 4096;
    auto timed = [&](const char* in, char* out) {
        float k = median_ms(7, [&] { FL(fl_u32_unpack(7, (const uint32_t*)in, (uint32_t*)out, n_blocks, nullptr)); });
        return (ib + ob) / k / 1e6;
    };
    auto line = [&](const char* what, const char* in, char* out) {
        FL(fl_fill_random((void*)in, ib, 11, nullptr));
        const double g = timed(in, out);
        printf("  %-64s in %p out %p (out - in = %+.4f GiB): %5.0f (%.3f)\n", what, (const void*)in, (void*)out, ((double)(out - in)) / GiB, g, g / 8000);
        fflush(stdout);
    };
    for (int rep = 0; rep < 3; ++rep) {
        void *a = nullptr, *b = nullptr;
        if (rep == 1) { CK(hipMalloc(&b, ob)); CK(hipMalloc(&a, ib)); }
        else { CK(hipMalloc(&a, ib)); CK(hipMalloc(&b, ob)); }
        line(rep == 1 ? "two hipMallocs (output allocated first)" : "two hipMallocs (input allocated first)", (const char*)a, (char*)b);
        CK(hipFree(a));
        CK(hipFree(b));
    }
    {
        char* slab = nullptr;
        CK(hipMalloc((void**)&slab, 64 * GiB));
        printf("one hipMalloc of 64 GiB, input at its start, output at <offset>:\n");
        for (size_t off_mib : {9216, 9216 + 2, 9216 + 64, 9216 + 512, 10240, 12288, 16384, 20480, 24576}) {
            char w[64];
            snprintf(w, sizeof w, "slab, out at %zu MiB", off_mib);
            line(w, slab, slab + off_mib * MiB);
        }
        CK(hipFree(slab));
    }
    Pool p;
    p.chunk = chunk_mib * MiB;
    const size_t nin = (ib + p.chunk - 1) / p.chunk, nout = (ob + p.chunk - 1) / p.chunk, n = nin + nout;
    p.h.resize(n);
    for (size_t i = 0; i < n; ++i) CK(hipMemCreate(&p.h[i], p.chunk, &prop, 0));
    std::vector<int> idx_in(nin), idx_out(nout);
    for (size_t i = 0; i < nin; ++i) idx_in[i] = (int)i;
    for (size_t i = 0; i < nout; ++i) idx_out[i] = (int)(nin + i);
    printf("VMM, %zu chunks of %zu MiB, always the same chunks for the input (first %zu) and the output:\n", n, chunk_mib, nin);
    // (1) one reservation, output at several gaps behind the input
    for (size_t gap_mib : {0, 2, 64, 512, 1024, 4096, 16384}) {
        char* va = nullptr;
        const size_t gap = gap_mib * MiB, tot = n * p.chunk + gap;
        CK(hipMemAddressReserve((void**)&va, tot, 2 * MiB, nullptr, 0));
        map_chunks(va, p, idx_in);
        map_chunks(va + nin * p.chunk + gap, p, idx_out);
        char w[64];
        snprintf(w, sizeof w, "one reservation, gap %zu MiB between input and output", gap_mib);
        line(w, va, va + nin * p.chunk + gap);
        CK(hipDeviceSynchronize());
        unmap_chunks(va, p, nin);
        unmap_chunks(va + nin * p.chunk + gap, p, nout);
        CK(hipMemAddressFree(va, tot));
    }
    // (2) two reservations, in either order
    for (int order = 0; order < 2; ++order) {
        char *vi = nullptr, *vo = nullptr;
        if (order == 0) { CK(hipMemAddressReserve((void**)&vi, nin * p.chunk, 2 * MiB, nullptr, 0)); CK(hipMemAddressReserve((void**)&vo, nout * p.chunk, 2 * MiB, nullptr, 0)); }
        else { CK(hipMemAddressReserve((void**)&vo, nout * p.chunk, 2 * MiB, nullptr, 0)); CK(hipMemAddressReserve((void**)&vi, nin * p.chunk, 2 * MiB, nullptr, 0)); }
        map_chunks(vi, p, idx_in);
        map_chunks(vo, p, idx_out);
        line(order == 0 ? "two reservations (input reserved first)" : "two reservations (output reserved first)", vi, vo);
        CK(hipDeviceSynchronize());
        unmap_chunks(vi, p, nin);
        unmap_chunks(vo, p, nout);
        CK(hipMemAddressFree(vi, nin * p.chunk));
        CK(hipMemAddressFree(vo, nout * p.chunk));
    }
    // (3) output BEFORE the input in one reservation
    {
        char* va = nullptr;
        CK(hipMemAddressReserve((void**)&va, n * p.chunk, 2 * MiB, nullptr, 0));
        map_chunks(va, p, idx_out);
        map_chunks(va + nout * p.chunk, p, idx_in);
        line("one reservation, output first, input behind it", va + nout * p.chunk, va);
        CK(hipDeviceSynchronize());
        unmap_chunks(va, p, n);
        CK(hipMemAddressFree(va, n * p.chunk));
    }
    for (size_t i = 0; i < n; ++i) CK(hipMemRelease(p.h[i]));
}

int main(int argc, char** argv)
{
    if (argc > 1 && !strcmp(argv[1], "position")) {          // exp_vmm position <pool_GiB> <n_blocks> <chunk_MiB> [<chunk_MiB> ...]
        CK(hipSetDevice(DEV));
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = DEV;
        CK(hipEventCreate(&E0));
        CK(hipEventCreate(&E1));
        std::vector<size_t> cms;
        for (int i = 4; i < argc; ++i) cms.push_back(atol(argv[i]));
        position_mode(atol(argv[2]), cms, atol(argv[3]), prop);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "remap")) {
        CK(hipSetDevice(DEV));
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = DEV;
        remap_mode(prop);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "state")) {             // exp_vmm state <n_blocks>
        CK(hipSetDevice(DEV));
        CK(hipEventCreate(&E0));
        CK(hipEventCreate(&E1));
        state_mode(atol(argv[2]));
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "va")) {                // exp_vmm va <n_blocks> <chunk_MiB>
        CK(hipSetDevice(DEV));
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = DEV;
        CK(hipEventCreate(&E0));
        CK(hipEventCreate(&E1));
        va_mode(atol(argv[2]), atol(argv[3]), prop);
        return 0;
    }
    const size_t pool_gib = argc > 1 ? atol(argv[1]) : 144;
    const size_t chunk = (argc > 2 ? atol(argv[2]) : 1024) * MiB;
    const std::string workload = argc > 3 ? argv[3] : "unpack32w7";
    const size_t n_blocks = argc > 4 ? atol(argv[4]) : 10000000;
    const int rounds = argc > 5 ? atoi(argv[5]) : 2;
    const size_t va_align = (argc > 6 ? atol(argv[6]) : 2) * MiB;       // alignment of the address reservations
    const size_t slab_gib = argc > 7 ? atol(argv[7]) : 0;                // > 0: first the same experiment inside ONE hipMalloc of that size
    CK(hipSetDevice(DEV));
    int vmm = 0;
    CK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, DEV));
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = DEV;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    printf("VMM supported %d, granularity min %zu recommended %zu bytes; device memory free %.1f / %.1f GiB\n", vmm, gmin, grec,
           free_b / (double)GiB, total_b / (double)GiB);
    if (!vmm) return 0;
    CK(hipEventCreate(&E0));
    CK(hipEventCreate(&E1));


    // ---- 0. (optional) the same question inside ONE plain allocation: input at its start, output at every 4-GiB offset ----------------
    if (slab_gib) {
        char* slab = nullptr;
        CK(hipMalloc((void**)&slab, slab_gib * GiB));
        FL(fl_fill_random(slab, slab_gib * GiB, 3, nullptr));
        CK(hipDeviceSynchronize());
        std::vector<int> cls;
        printf("one hipMalloc of %zu GiB at %p, class of every GiB:\n", slab_gib, (void*)slab);
        classify(slab, slab_gib, GiB, cls);
        print_classes(cls);
        const size_t ib = n_blocks * 128 * 7, ob = n_blocks * 4096;
        const size_t first = (ib + 4 * GiB - 1) / (4 * GiB) * 4;
        FL(fl_fill_random(slab, ib, 11, nullptr));
        printf("unpack u32 W=7, %zu blocks, input at 0; output at <GiB>: kernel GB/s (of 8 TB/s)  [class of every 2nd GiB it covers]\n", n_blocks);
        for (size_t off = first; (off * GiB + ob) <= slab_gib * GiB; off += 4) {
            float k = median_ms(7, [&] { FL(fl_u32_unpack(7, (const uint32_t*)slab, (uint32_t*)(slab + off * GiB), n_blocks, nullptr)); });
            printf("  out@%3zu: %5.0f (%.3f)  [", off, (ib + ob) / k / 1e6, (ib + ob) / k / 8e9);
            for (size_t g = off; g * GiB < off * GiB + ob; g += 2) putchar(cls[g] < 0 ? '?' : 'A' + cls[g]);
            printf("]\n");
        }
        CK(hipFree(slab));
        fflush(stdout);
    }

    if (const char* pre = getenv("EXP_PREALLOC")) {            // "<GiB>[,touch][,keep]": a plain allocation made (and freed) before anything else
        const size_t g = atol(pre);
        void* q = nullptr;
        CK(hipMalloc(&q, g * GiB));
        if (strstr(pre, "touch")) { FL(fl_fill_random(q, g * GiB, 5, nullptr)); CK(hipDeviceSynchronize()); }
        if (!strstr(pre, "keep")) CK(hipFree(q));
        printf("before the pool: hipMalloc of %zu GiB%s%s\n", g, strstr(pre, "touch") ? ", filled" : "", strstr(pre, "keep") ? ", kept" : ", freed");
    }
    // ---- 1. the pool ------------------------------------------------------------------------------------------------------------
    Pool p;
    p.chunk = chunk;
    const size_t n_chunks = pool_gib * GiB / chunk;
    p.h.resize(n_chunks);
    p.cls.assign(n_chunks, -1);
    CK(hipEventRecord(E0, nullptr));
    for (size_t i = 0; i < n_chunks; ++i) CK(hipMemCreate(&p.h[i], chunk, &prop, 0));
    char* va = nullptr;
    CK(hipMemAddressReserve((void**)&va, n_chunks * chunk, va_align, nullptr, 0));
    std::vector<int> ident(n_chunks);
    for (size_t i = 0; i < n_chunks; ++i) ident[i] = (int)i;
    map_chunks(va, p, ident);
    CK(hipEventRecord(E1, nullptr));
    CK(hipEventSynchronize(E1));
    float t_create;
    CK(hipEventElapsedTime(&t_create, E0, E1));
    printf("pool: %zu chunks of %zu MiB created + mapped at %p in %.1f ms\n", n_chunks, chunk / MiB, (void*)va, t_create);
    FL(fl_fill_random(va, n_chunks * chunk, 3, nullptr));
    CK(hipDeviceSynchronize());

    const int n_classes = classify(va, n_chunks, chunk, p.cls);
    printf("class of every chunk in creation order ('?' = none):\n");
    print_classes(p.cls);
    std::vector<std::vector<int>> by_class(n_classes + 1);
    for (size_t g = 0; g < n_chunks; ++g) by_class[p.cls[g] < 0 ? n_classes : p.cls[g]].push_back((int)g);
    if (getenv("EXP_LAYOUTS")) {                               // letters by plenty: 'A' = the class with the most chunks (a layout's big side fits)
        std::vector<int> perm(n_classes);
        for (int c = 0; c < n_classes; ++c) perm[c] = c;
        std::sort(perm.begin(), perm.end(), [&](int a, int b) { return by_class[a].size() > by_class[b].size(); });
        std::vector<std::vector<int>> sorted(n_classes + 1);
        for (int c = 0; c < n_classes; ++c) sorted[c] = by_class[perm[c]];
        sorted[n_classes] = by_class[n_classes];
        by_class.swap(sorted);
        for (int c = 0; c < n_classes; ++c) for (int g : by_class[c]) p.cls[g] = c;
        printf("classes relabelled by plenty:\n");
        print_classes(p.cls);
    }
    for (int c = 0; c < n_classes; ++c) printf("class %c: %zu chunks; ", 'A' + c, by_class[c].size());
    printf("unclassified: %zu\n", by_class[n_classes].size());
    unmap_chunks(va, p, n_chunks);
    const bool keep_va = getenv("EXP_KEEP_VA") != nullptr, sep_late = getenv("EXP_SEP_LATE") != nullptr;
    if (!keep_va) CK(hipMemAddressFree(va, n_chunks * chunk));
    printf("pool address range %s; the two hipMallocs are made %s the layouts\n", keep_va ? "kept reserved" : "freed", sep_late ? "after" : "before");
    fflush(stdout);

    // ---- 3. layouts --------------------------------------------------------------------------------------------------------------
    size_t in_bytes, out_bytes;
    unsigned W = 7;
    bool mixed = false, pack = false;
    bool transpose = false;
    if (workload.rfind("unpack32w", 0) == 0) { W = (unsigned)atoi(workload.c_str() + 9); in_bytes = n_blocks * 128 * W; out_bytes = n_blocks * 4096; }
    else if (workload.rfind("pack32w", 0) == 0) { pack = true; W = (unsigned)atoi(workload.c_str() + 7); in_bytes = n_blocks * 4096; out_bytes = n_blocks * 128 * W; }
    else if (workload == "transpose32") { transpose = true; W = 32; in_bytes = out_bytes = n_blocks * 4096; }
    else if (workload == "mixed32") { mixed = true; in_bytes = 0; out_bytes = n_blocks * 4096; }
    else { printf("unknown workload\n"); return 1; }
    uint8_t* d_widths = nullptr;
    uint64_t *d_offsets = nullptr, *d_total = nullptr;
    uint32_t* d_err = nullptr;
    if (mixed) {
        std::vector<uint8_t> w(n_blocks);
        for (size_t b = 0; b < n_blocks; ++b) { w[b] = 1 + b % 32; in_bytes += 128u * w[b]; }
        CK(hipMalloc((void**)&d_widths, n_blocks));
        CK(hipMalloc((void**)&d_offsets, n_blocks * 8));
        CK(hipMalloc((void**)&d_total, 8));
        CK(hipMalloc((void**)&d_err, 4));
        CK(hipMemset(d_err, 0, 4));
        CK(hipMemcpy(d_widths, w.data(), n_blocks, hipMemcpyHostToDevice));
        FL(fl_widths_to_offsets(32, d_widths, n_blocks, d_offsets, d_total, d_err, nullptr));
        CK(hipDeviceSynchronize());
    }
    const size_t n_in = (in_bytes + chunk - 1) / chunk, n_out = (out_bytes + chunk - 1) / chunk;
    printf("workload %s, %zu blocks: in %.2f GiB (%zu chunks), out %.2f GiB (%zu chunks)\n", workload.c_str(), n_blocks,
           in_bytes / (double)GiB, n_in, out_bytes / (double)GiB, n_out);
    char *va_in = nullptr, *va_out = nullptr;
    const bool one_range = getenv("EXP_ONE_RANGE") != nullptr, no_stream = getenv("EXP_NO_STREAM") != nullptr;
    if (one_range) {
        CK(hipMemAddressReserve((void**)&va_in, (n_in + n_out) * chunk, va_align, nullptr, 0));
        va_out = va_in + n_in * chunk;
    } else {
        CK(hipMemAddressReserve((void**)&va_in, n_in * chunk, va_align, nullptr, 0));
        CK(hipMemAddressReserve((void**)&va_out, n_out * chunk, va_align, nullptr, 0));
    }
    printf("one range %d, bare stream skipped %d\n", (int)one_range, (int)no_stream);
    printf("address ranges: in %p, out %p (alignment asked %zu MiB)\n", (void*)va_in, (void*)va_out, va_align / MiB);

    size_t in_unit, aux_unit, out_unit;
    int nt, waves, win;
    unsigned bpu;
    FL(fl_internal_bare_stream_shape(mixed ? 3 : pack ? 1 : 0, 32, mixed ? 33 : W, &in_unit, &aux_unit, &out_unit, &nt, &waves, &win, &bpu));
    auto run = [&](const void* in, void* out, double& k_gbps, double& s_gbps) {
        const double bytes = (double)in_bytes + out_bytes;
        float k = median_ms(9, [&] {
            if (mixed) FL(fl_u32_unpack_widths(d_widths, d_offsets, (const uint32_t*)in, in_bytes, (uint32_t*)out, n_blocks, d_err, nullptr));
            else if (transpose) FL(fl_u32_transpose((const uint32_t*)in, (uint32_t*)out, n_blocks, nullptr));
            else if (pack) FL(fl_u32_pack(W, (const uint32_t*)in, (uint32_t*)out, n_blocks, nullptr));
            else FL(fl_u32_unpack(W, (const uint32_t*)in, (uint32_t*)out, n_blocks, nullptr));
        });
        float s = no_stream ? 1.f : median_ms(5, [&] { FL(fl_internal_bare_stream(in, in_unit, nullptr, 0, out, out_unit, n_blocks / bpu, nt, waves, win, nullptr)); });
        k_gbps = bytes / k / 1e6;
        s_gbps = (double)(n_blocks / bpu) * (in_unit + out_unit) / s / 1e6;
    };

    // a layout = class pattern of the input chunks + class pattern of the output chunks + run length (chunks per letter)
    struct Layout { std::string name, in_pat, out_pat; int run; int in_run = 1; };
    if (const char* pol = getenv("EXP_POLICY")) {              // fl_internal_set_kernel_policy for every launch (e.g. 31 << 25 = whole-column tile map)
        fl_internal_set_kernel_policy(atoi(pol));
        printf("kernel policy %d (fastlanes_amd_internal.h)\n", fl_internal_get_kernel_policy());
    }
    std::vector<Layout> layouts = {
        {"in A   | out A", "A", "A", 1},       {"in A   | out B", "A", "B", 1},         {"in A   | out AB/1", "A", "AB", 1},
        {"in A   | out BC/1", "A", "BC", 1},   {"in A   | out ABC/1", "A", "ABC", 1},   {"in A   | out BC/4", "A", "BC", 4},
        {"in A   | out B then C", "A", "BC", -1}, {"in A   | out AB/4", "A", "AB", 4},  {"in ABC | out ABC/1", "ABC", "ABC", 1},
        {"in BC  | out BC/1", "BC", "BC", 1},  {"in A   | out BCB then CBC eighths", "A", "BC", -8},
        {"in A   | out B then A", "A", "BA", -1}, {"in A   | out ABC/2", "A", "ABC", 2}, {"in A   | out ABC thirds", "A", "ABC", -3},
    };
    if (const char* spec = getenv("EXP_LAYOUTS")) {            // "inpat,inrun,outpat,outrun;..." (run > 0: chunks per letter; < 0: |run| equal stretches)
        layouts.clear();
        std::string all = spec;
        size_t at = 0;
        while (at < all.size()) {
            size_t end = all.find(';', at);
            if (end == std::string::npos) end = all.size();
            const std::string one = all.substr(at, end - at);
            char ip[32] = {0}, op[32] = {0};
            int ir = 1, orun = 1;
            if (sscanf(one.c_str(), "%31[A-C],%d,%31[A-C],%d", ip, &ir, op, &orun) == 4 && ir && orun) {
                Layout L{"in " + std::string(ip) + "/" + std::to_string(ir) + " | out " + op + "/" + std::to_string(orun), ip, op, orun};
                L.in_run = ir;
                layouts.push_back(L);
            } else printf("EXP_LAYOUTS: cannot read '%s'\n", one.c_str());
            at = end + 1;
        }
    }
    struct Result { std::vector<double> k, s; bool ok = true; };
    std::vector<Result> res(layouts.size() + 2);
    // take chunks of the asked class round-robin from per-class free lists; run < 0: |run| equal stretches over the buffer
    auto build = [&](const std::string& pat, int run, size_t n, std::vector<size_t>& next, std::vector<int>& idx) -> bool {
        idx.clear();
        for (size_t i = 0; i < n; ++i) {
            // run 99 = by POSITION under the whole-column tile map: XCD x walks the x-th eighth of the buffer, all eight at the same pace, so
            // chunk i = the k-th chunk of eighth x gets letter (x + k): at any moment the eight write positions cycle through the letters
            size_t step = run > 0 ? i / run : i * (size_t)(-run) / n;
            if (run == 99) {
                const size_t x = i * 8 / n, first = (x * n + 7) / 8;
                step = x + (i >= first ? i - first : 0);
            }
            const int c = pat[step % pat.size()] - 'A';
            if (c >= n_classes || next[c] >= by_class[c].size()) return false;
            idx.push_back(by_class[c][next[c]++]);
        }
        return true;
    };
    void *sep_in = nullptr, *sep_out = nullptr;
    if (!sep_late) { CK(hipMalloc(&sep_in, in_bytes)); CK(hipMalloc(&sep_out, out_bytes)); }
    auto fill_in = [&](void* in) {
        if (pack) {                                             // W-bit values
            FL(fl_fill_random(in, in_bytes, 11, nullptr));
        } else FL(fl_fill_random(in, in_bytes, 11, nullptr));
    };
    if (!sep_late) fill_in(sep_in);
    if (const char* warm = getenv("EXP_WARM")) {              // N seconds of back-to-back unpack launches on the creation-order mapping first
        std::vector<int> ii(n_in), io(n_out);
        for (size_t i = 0; i < n_in; ++i) ii[i] = (int)i;
        for (size_t i = 0; i < n_out; ++i) io[i] = (int)(n_in + i);
        char *w_in = va_in, *w_out = va_out;
        const bool elsewhere = getenv("EXP_WARM_ELSEWHERE") != nullptr;
        if (elsewhere) {
            CK(hipMemAddressReserve((void**)&w_in, n_in * chunk, va_align, nullptr, 0));
            CK(hipMemAddressReserve((void**)&w_out, n_out * chunk, va_align, nullptr, 0));
            printf("warm-up on its own address ranges %p / %p\n", (void*)w_in, (void*)w_out);
        }
        map_chunks(w_in, p, ii);
        map_chunks(w_out, p, io);
        fill_in(w_in);
        const int launches = (int)(atof(warm) * 1000 / 7.5);
        printf("warm-up: %d back-to-back launches on the creation-order mapping; every 40th launch's ms:", launches);
        for (int i = 0; i < launches; i += 40) {
            float t = median_ms(launches < 39 ? launches : 39, [&] { FL(fl_u32_unpack(W, (const uint32_t*)w_in, (uint32_t*)w_out, n_blocks, nullptr)); });
            printf(" %.2f", t);
        }
        printf("\n");
        CK(hipDeviceSynchronize());
        unmap_chunks(w_in, p, n_in);
        unmap_chunks(w_out, p, n_out);
    }
    for (int r = 0; r < rounds; ++r) {
        for (size_t li = 0; li < layouts.size(); ++li) {
            const Layout& L = layouts[li];
            std::vector<size_t> next(n_classes, 0);
            std::vector<int> idx_in, idx_out;
            if (!build(L.in_pat, L.in_run, n_in, next, idx_in) || !build(L.out_pat, L.run, n_out, next, idx_out)) { res[li].ok = false; continue; }
            // a FRESH pair of address ranges for every layout: on this ROCm (7.2) hipMemUnmap + hipMemMap of OTHER chunks at an address that
            // was mapped before leaves the device reading and writing the FIRST chunks (exp_vmm remap; profiles/r06_vmm_remap.txt)
            CK(hipMemAddressReserve((void**)&va_in, n_in * chunk, va_align, nullptr, 0));
            CK(hipMemAddressReserve((void**)&va_out, n_out * chunk, va_align, nullptr, 0));
            map_chunks(va_in, p, idx_in);
            map_chunks(va_out, p, idx_out);
            fill_in(va_in);
            double k, s;
            run(va_in, va_out, k, s);
            res[li].k.push_back(k);
            res[li].s.push_back(s);
            CK(hipDeviceSynchronize());
            unmap_chunks(va_in, p, n_in);
            unmap_chunks(va_out, p, n_out);
        }
        {   // creation order, whatever classes that gives (what a plain allocation through this API would be)
            std::vector<int> idx_in(n_in), idx_out(n_out);
            for (size_t i = 0; i < n_in; ++i) idx_in[i] = (int)i;
            for (size_t i = 0; i < n_out; ++i) idx_out[i] = (int)(n_in + i);
            CK(hipMemAddressReserve((void**)&va_in, n_in * chunk, va_align, nullptr, 0));
            CK(hipMemAddressReserve((void**)&va_out, n_out * chunk, va_align, nullptr, 0));
            map_chunks(va_in, p, idx_in);
            map_chunks(va_out, p, idx_out);
            fill_in(va_in);
            double k, s;
            run(va_in, va_out, k, s);
            res[layouts.size()].k.push_back(k);
            res[layouts.size()].s.push_back(s);
            CK(hipDeviceSynchronize());
            unmap_chunks(va_in, p, n_in);
            unmap_chunks(va_out, p, n_out);
        }
        if (sep_late && !sep_in) { CK(hipMalloc(&sep_in, in_bytes)); CK(hipMalloc(&sep_out, out_bytes)); fill_in(sep_in); }
        double k, s;
        run(sep_in, sep_out, k, s);
        res[layouts.size() + 1].k.push_back(k);
        res[layouts.size() + 1].s.push_back(s);
    }
    if (getenv("EXP_FRESH_VA_AFTER")) {
        char *f_in = nullptr, *f_out = nullptr;
        CK(hipMemAddressReserve((void**)&f_in, n_in * chunk, va_align, nullptr, 0));
        CK(hipMemAddressReserve((void**)&f_out, n_out * chunk, va_align, nullptr, 0));
        for (int variant = 0; variant < 2; ++variant) {
            std::vector<int> ii(n_in), io(n_out);
            for (size_t i = 0; i < n_in; ++i) ii[i] = (int)(variant ? n_out + i : i);
            for (size_t i = 0; i < n_out; ++i) io[i] = (int)(variant ? i : n_in + i);
            map_chunks(f_in, p, ii);
            map_chunks(f_out, p, io);
            fill_in(f_in);
            double k, s2;
            run(f_in, f_out, k, s2);
            printf("after the layouts, FRESH address ranges %p / %p, creation order%s: %5.0f (%.3f)\n", (void*)f_in, (void*)f_out, variant ? " (output = the first chunks)" : "", k, k / 8000);
            CK(hipDeviceSynchronize());
            unmap_chunks(f_in, p, n_in);
            unmap_chunks(f_out, p, n_out);
        }
    }
    if (getenv("EXP_VA_SWEEP")) {
        // the SAME chunks (creation order) at addresses that differ only in their alignment: base = a 1-GiB multiple + delta
        const size_t span = (n_in + n_out) * chunk, extra = 3 * GiB;
        char* big = nullptr;
        CK(hipMemAddressReserve((void**)&big, span + extra, 2 * MiB, nullptr, 0));
        char* aligned = (char*)(((uintptr_t)big + GiB - 1) & ~(uintptr_t)(GiB - 1));
        std::vector<int> idx(n_in + n_out);
        for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int)i;
        printf("the first %zu chunks in creation order, mapped at <1-GiB multiple %p> + delta:\n", idx.size(), (void*)aligned);
        for (size_t d_mib : {0, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 1026, 0}) {
            char* base = aligned + d_mib * MiB;
            map_chunks(base, p, idx);
            fill_in(base);
            double k, s2;
            run(base, base + n_in * chunk, k, s2);
            printf("  delta %5zu MiB: %5.0f (%.3f)\n", d_mib, k, k / 8000);
            fflush(stdout);
            CK(hipDeviceSynchronize());
            unmap_chunks(base, p, idx.size());
        }
        CK(hipMemAddressFree(big, span + extra));
        fflush(stdout);
    }
    if (getenv("EXP_LIBRARY_PAIR")) {                           // the product's own constructed pair, next to this tool's layouts
        for (int rep = 0; rep < 2; ++rep) {
            void *li = nullptr, *la = nullptr, *lo = nullptr, *lh = nullptr;
            int kept = -1;
            FL(fl_column_pair_alloc(in_bytes, 0, out_bytes, FL_LAYOUT_INTERLEAVED, nullptr, &li, &la, &lo, &lh, &kept, nullptr));
            fill_in(li);
            double k, s2;
            run(li, lo, k, s2);
            printf("fl_column_pair_alloc(FL_LAYOUT_INTERLEAVED), pool still held by this tool: %5.0f (%.3f) stream %5.0f  classes %s\n", k, k / 8000, s2,
                   fl_internal_column_pair_classes(lh));
            CK(hipDeviceSynchronize());
            FL(fl_column_pair_free(lh));
        }
        fflush(stdout);
    }
    printf("%-40s %s\n", "layout (classes by chunk; /n = run length)", "kernel GB/s (of 8 TB/s) per round | bare stream GB/s per round");
    for (size_t li = 0; li < res.size(); ++li) {
        const char* name = li < layouts.size() ? layouts[li].name.c_str() : li == layouts.size() ? "creation order (VMM, no choice)" : "two hipMallocs";
        printf("%-40s", name);
        if (!res[li].ok) { printf(" (not enough chunks of a class)\n"); continue; }
        for (double k : res[li].k) printf(" %5.0f (%.3f)", k, k / 8000.0);
        printf(" |");
        for (double s : res[li].s) printf(" %5.0f", s);
        printf("\n");
    }
    if (mixed) {
        uint32_t err = 0;
        CK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
        printf("device error flag %u\n", err);
    }
    return 0;
}
