/* multi_gpu_decode.c -- a sharded column decoded on every GPU of the node through the C ABI alone: one process, one host
 * thread per device (its own hipSetDevice + stream), no torch, no RCCL, no collective of any kind.
 *
 * Every 1024-value block is independent in every codec function (bitpacking.rs:19,33 take one block; Delta's bases are
 * per block, delta.rs:7), so the reference's caller loop
 *
 *     for b in 0..n { u32::unchecked_unpack(width[b], &packed[off[b]..], &mut out[b*1024..]) }      (bitpacking.rs:109-129,
 *                                                                                 loop shape benches/bitpacking.rs:80-97)
 *
 * shards by contiguous block range: device g owns blocks [g*N/G, (g+1)*N/G) and its slice of the packed bytes.
 * Two legs, the two multi-GPU shapes of BASELINE.json:
 *   weak    configs[1] on every device: u32 W=7 unpack, --blocks blocks per device          (fl_u32_unpack)
 *   strong  configs[4]: the 10 B-integer column (9 765 625 blocks, width[b] = 1 + b mod 32) split over the devices,
 *           widths[] / offsets[] device-resident                                            (fl_u32_unpack_widths)
 * Per device: kernel GB/s (HIP events on the thread's own stream); aggregate: bytes of all devices / wall time of the
 * slowest thread between two barriers.  Each thread verifies the first, the last and a few sampled blocks of ITS slice
 * against a scalar decoder written from the wire format (unpack_single's closed form, bitpacking.rs:132-179,207-232).
 *
 * Build: gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/multi_gpu_decode.c \
 *            -L fastlanes_amd -lfastlanes_amd -L /opt/rocm/lib -lamdhip64 -lpthread -Wl,-rpath,'$ORIGIN/../fastlanes_amd' \
 *            -o examples/multi_gpu_decode
 * Run:   examples/multi_gpu_decode [--blocks N] [--strong-blocks N] [--steps K] [--warmup W] [--devices D] [--replicas R]
 *        --replicas R puts R threads (each with its own stream and slice) on every device: the N-thread path on a 1-GPU box.
 * Prints one JSON line; exit code 0 iff every slice verified.
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <hip/hip_runtime_api.h>
#include "fastlanes_amd.h"

#define MAX_THREADS 64
/* Where a column lives in HBM moves the same kernel by a few per cent; nothing in an address tells which
 * layout is the fast one on a given box, and the library takes any 16-byte aligned pointers.  Default: two plain hipMalloc's.
 * --zoned carves each leg's buffers from ONE allocation instead -- the input at 0, the output centred on the slab's 64-GiB offset
 * (the other layout bench.py's --placement auto times; fastlanes_amd/placement.py). */
#define ZONE_BYTES ((size_t)64 << 30)
static int g_zoned = 0;

/* two separate allocations if the slab does not fit (or the input would run into the output) */
static int alloc_pair(size_t in_bytes, size_t out_bytes, void **slab, void **in, void **out)
{
    const size_t out_off = (ZONE_BYTES - out_bytes / 2) & ~(size_t)255;
    *slab = NULL;
    if (g_zoned && out_bytes / 2 <= ZONE_BYTES && in_bytes <= out_off && hipMalloc(slab, out_off + (out_bytes ? out_bytes : 16)) == hipSuccess) {
        *in = *slab;
        *out = (char *)*slab + out_off;
        return 0;
    }
    (void)hipGetLastError();
    *slab = NULL;
    if (hipMalloc(in, in_bytes ? in_bytes : 16) != hipSuccess) return 1;
    if (hipMalloc(out, out_bytes ? out_bytes : 16) != hipSuccess) return 1;
    return 0;
}
static void free_pair(void *slab, void *in, void *out)
{
    if (slab) { (void)hipFree(slab); return; }
    if (in) (void)hipFree(in);
    if (out) (void)hipFree(out);
}

typedef struct {
    int tid, n_threads, device, steps, warmup;
    size_t weak_blocks;                   /* blocks of the weak leg on this thread */
    size_t strong_first, strong_blocks;   /* this thread's range of the strong column */
    pthread_barrier_t *bar;
    /* results */
    int status;                           /* 0 ok, 1 mismatch, 2 HIP / library error */
    char err[160];
    double weak_kernel_ms, strong_kernel_ms, weak_bytes, strong_bytes;
    double weak_wall_s, strong_wall_s;    /* between the two barriers, measured by this thread */
} worker_t;

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* value i (0..1023) of one packed u32 block of width w: the closed form of bitpacking.rs:132-179 with the index tables
 * of :207-232 (lane = i % 32; row = FL_ORDER[(i - s*128 - lane) / 16] * 8 + s, s = i / 128) */
static uint32_t ref_unpack_single_u32(unsigned w, const uint32_t *pk, unsigned i)
{
    static const unsigned FL_ORDER[8] = {0, 4, 2, 6, 1, 5, 3, 7};            /* lib.rs:22 */
    if (w == 0) return 0;
    const unsigned lane = i % 32, s = i / 128, row = FL_ORDER[(i - s * 128 - lane) / 16] * 8 + s;
    if (w == 32) return pk[32 * row + lane];
    const unsigned start = row * w, word = start / 32, sh = start % 32;
    uint32_t v = pk[32 * word + lane] >> sh;
    if (32 - sh < w) v |= pk[32 * (word + 1) + lane] << (32 - sh);
    return v & ((1u << w) - 1u);
}

#define HIP_OR_FAIL(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    snprintf(wk->err, sizeof wk->err, "HIP error %d (%s) at line %d", (int)e_, hipGetErrorString(e_), __LINE__); wk->status = 2; goto done; } } while (0)
#define FL_OR_FAIL(x) do { int s_ = (x); if (s_ != FL_OK) { \
    snprintf(wk->err, sizeof wk->err, "%s at line %d", fl_status_string(s_), __LINE__); wk->status = 2; goto done; } } while (0)

/* compare block b of a decoded slice with the scalar decode of its packed bytes (both copied back from the device) */
static int verify_block(worker_t *wk, const uint32_t *d_packed, uint64_t byte_off, unsigned w, const uint32_t *d_out, size_t b)
{
    uint32_t pk[32 * 32], out[1024];
    if (w && hipMemcpy(pk, (const char *)d_packed + byte_off, 128u * w, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    if (hipMemcpy(out, d_out + b * 1024, sizeof out, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    for (unsigned i = 0; i < 1024; ++i)
        if (out[i] != ref_unpack_single_u32(w, pk, i)) {
            snprintf(wk->err, sizeof wk->err, "thread %d: block %zu value %u differs from the scalar decode", wk->tid, b, i);
            return 1;
        }
    return 0;
}

/* K launches of one leg between two barriers; kernel time from HIP events on this thread's stream */
typedef int (*launch_fn)(void *ctx, hipStream_t st);

static int timed_leg(worker_t *wk, hipStream_t st, launch_fn launch, void *ctx, double *kernel_ms, double *wall_s)
{
    /* ALWAYS passes both barriers, whatever fails: the other threads wait in them (the caller counts two per call) */
    hipEvent_t e0 = NULL, e1 = NULL;
    int rc = FL_OK;
    if (hipEventCreate(&e0) != hipSuccess) { e0 = NULL; rc = FL_ERR_HIP; }
    if (rc == FL_OK && hipEventCreate(&e1) != hipSuccess) { e1 = NULL; rc = FL_ERR_HIP; }
    for (int i = 0; i < wk->warmup && rc == FL_OK; ++i) rc = launch(ctx, st);
    if (hipStreamSynchronize(st) != hipSuccess) rc = FL_ERR_HIP;
    pthread_barrier_wait(wk->bar);                 /* every device starts its K steps together ... */
    const double t0 = now_s();
    if (rc == FL_OK && hipEventRecord(e0, st) != hipSuccess) rc = FL_ERR_HIP;
    for (int i = 0; i < wk->steps && rc == FL_OK; ++i) rc = launch(ctx, st);
    if (rc == FL_OK && hipEventRecord(e1, st) != hipSuccess) rc = FL_ERR_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) rc = FL_ERR_HIP;
    *wall_s = now_s() - t0;
    pthread_barrier_wait(wk->bar);                 /* ... and the leg ends when the slowest one is done */
    float ms = 0.f;
    if (rc == FL_OK && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = FL_ERR_HIP;
    *kernel_ms = (double)ms / (wk->steps > 0 ? wk->steps : 1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return rc;
}

typedef struct { const uint32_t *packed; uint32_t *out; size_t n; } weak_ctx;
static int weak_launch(void *c, hipStream_t st)
{
    weak_ctx *x = c;
    return fl_u32_unpack(7, x->packed, x->out, x->n, st);
}
typedef struct { const uint8_t *w; const uint64_t *off; const uint32_t *packed; size_t pbytes; uint32_t *out; size_t n; uint32_t *err; } strong_ctx;
static int strong_launch(void *c, hipStream_t st)
{
    strong_ctx *x = c;
    return fl_u32_unpack_widths(x->w, x->off, x->packed, x->pbytes, x->out, x->n, x->err, st);
}

static void *worker(void *arg)
{
    worker_t *wk = arg;
    hipStream_t st = NULL;
    uint32_t *d_pk = NULL, *d_out = NULL, *d_err = NULL;
    void *slab = NULL;
    uint8_t *d_w = NULL, *h_w = NULL;
    uint64_t *d_off = NULL, *d_total = NULL;
    int waited = 0;                                /* barriers passed so far (4 in a clean run) */
    wk->status = 0;
    HIP_OR_FAIL(hipSetDevice(wk->device));         /* per-thread: every call below targets this thread's device */
    HIP_OR_FAIL(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));

    /* ---- weak leg: BASELINE.json configs[1] on every device --------------------------------------------------- */
    {
        const size_t n = wk->weak_blocks, pbytes = n * 896, obytes = n * 4096;
        if (alloc_pair(pbytes, obytes, &slab, (void **)&d_pk, (void **)&d_out)) { wk->status = 2; snprintf(wk->err, sizeof wk->err, "out of device memory"); goto done; }
        FL_OR_FAIL(fl_fill_random(d_pk, pbytes, 1234u + (uint64_t)wk->tid, st));     /* never constant data: DVFS */
        weak_ctx c = {d_pk, d_out, n};
        int rc = timed_leg(wk, st, weak_launch, &c, &wk->weak_kernel_ms, &wk->weak_wall_s);
        waited += 2;
        FL_OR_FAIL(rc);
        wk->weak_bytes = (double)(pbytes + obytes);
        const size_t probe[5] = {0, n / 3, n / 2, n > 1 ? n - 2 : 0, n ? n - 1 : 0};
        for (int k = 0; k < 5 && n; ++k) {
            const int v = verify_block(wk, d_pk, (uint64_t)probe[k] * 896, 7, d_out, probe[k]);
            if (v) { wk->status = v; if (v == 2) snprintf(wk->err, sizeof wk->err, "copy-back failed"); goto done; }
        }
        free_pair(slab, d_pk, d_out);
        slab = NULL;
        d_pk = d_out = NULL;
    }
    /* ---- strong leg: BASELINE.json configs[4], this thread's contiguous block range ----------------------------- */
    {
        const size_t n = wk->strong_blocks;
        uint64_t total = 0;
        uint32_t err = 0;
        h_w = malloc(n ? n : 1);
        if (!h_w) { wk->status = 2; snprintf(wk->err, sizeof wk->err, "out of host memory"); goto done; }
        for (size_t b = 0; b < n; ++b) h_w[b] = (uint8_t)(1 + (wk->strong_first + b) % 32);
        HIP_OR_FAIL(hipMalloc((void **)&d_w, n ? n : 16));
        HIP_OR_FAIL(hipMalloc((void **)&d_off, (n ? n : 1) * sizeof(uint64_t)));
        HIP_OR_FAIL(hipMalloc((void **)&d_total, sizeof(uint64_t)));
        HIP_OR_FAIL(hipMalloc((void **)&d_err, sizeof(uint32_t)));
        HIP_OR_FAIL(hipMemsetAsync(d_err, 0, sizeof(uint32_t), st));
        HIP_OR_FAIL(hipMemcpyAsync(d_w, h_w, n, hipMemcpyHostToDevice, st));
        FL_OR_FAIL(fl_widths_to_offsets(32, d_w, n, d_off, d_total, d_err, st));     /* offsets built on the device */
        HIP_OR_FAIL(hipMemcpyAsync(&total, d_total, sizeof total, hipMemcpyDeviceToHost, st));
        HIP_OR_FAIL(hipStreamSynchronize(st));
        if (alloc_pair(total, n * 4096, &slab, (void **)&d_pk, (void **)&d_out)) { wk->status = 2; snprintf(wk->err, sizeof wk->err, "out of device memory"); goto done; }
        FL_OR_FAIL(fl_fill_random(d_pk, total, 4321u + (uint64_t)wk->tid, st));
        strong_ctx c = {d_w, d_off, d_pk, total, d_out, n, d_err};
        int rc = timed_leg(wk, st, strong_launch, &c, &wk->strong_kernel_ms, &wk->strong_wall_s);
        waited += 2;
        FL_OR_FAIL(rc);
        wk->strong_bytes = (double)total + (double)n * 4096.0;
        HIP_OR_FAIL(hipMemcpy(&err, d_err, sizeof err, hipMemcpyDeviceToHost));
        if (err) { wk->status = 1; snprintf(wk->err, sizeof wk->err, "device error flag %u", err); goto done; }
        const size_t probe[6] = {0, 1, n / 3, n / 2, n > 1 ? n - 2 : 0, n ? n - 1 : 0};
        for (int k = 0; k < 6 && n; ++k) {
            uint64_t off = 0;
            HIP_OR_FAIL(hipMemcpy(&off, d_off + probe[k], sizeof off, hipMemcpyDeviceToHost));
            const int v = verify_block(wk, d_pk, off, h_w[probe[k]], d_out, probe[k]);
            if (v) { wk->status = v; if (v == 2) snprintf(wk->err, sizeof wk->err, "copy-back failed"); goto done; }
        }
    }
done:
    /* a thread that failed early still has to meet the others at the barriers it skipped */
    for (; waited < 4; ++waited) pthread_barrier_wait(wk->bar);
    free(h_w);
    free_pair(slab, d_pk, d_out);
    if (d_w) (void)hipFree(d_w);
    if (d_off) (void)hipFree(d_off);
    if (d_total) (void)hipFree(d_total);
    if (d_err) (void)hipFree(d_err);
    if (st) (void)hipStreamDestroy(st);
    return NULL;
}

static size_t arg_size(int argc, char **argv, const char *name, size_t dflt)
{
    for (int i = 1; i + 1 < argc; ++i)
        if (!strcmp(argv[i], name)) return (size_t)strtoull(argv[i + 1], NULL, 10);
    return dflt;
}

int main(int argc, char **argv)
{
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) {
        fprintf(stderr, "multi_gpu_decode: no GPU visible (there is no CPU path)\n");
        return 2;
    }
    const size_t want_dev = arg_size(argc, argv, "--devices", (size_t)n_dev);
    if (want_dev < (size_t)n_dev) n_dev = (int)want_dev;
    const int replicas = (int)arg_size(argc, argv, "--replicas", 1);
    const int n_threads = n_dev * (replicas < 1 ? 1 : replicas);
    if (n_threads > MAX_THREADS) { fprintf(stderr, "too many threads\n"); return 2; }
    const size_t weak_blocks = arg_size(argc, argv, "--blocks", 10000000);
    const size_t strong_total = arg_size(argc, argv, "--strong-blocks", 9765625);
    const int steps = (int)arg_size(argc, argv, "--steps", 10), warmup = (int)arg_size(argc, argv, "--warmup", 2);

    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)n_threads);
    worker_t wk[MAX_THREADS];
    pthread_t th[MAX_THREADS];
    memset(wk, 0, sizeof wk);
    for (int t = 0; t < n_threads; ++t) {
        /* contiguous block ranges; the first (total % threads) threads hold one block more (8 GPUs: 1 220 704 + 7 x 1 220 703) */
        const size_t base = strong_total / (size_t)n_threads, rem = strong_total % (size_t)n_threads;
        wk[t].tid = t; wk[t].n_threads = n_threads; wk[t].device = t % n_dev;
        wk[t].steps = steps; wk[t].warmup = warmup; wk[t].bar = &bar;
        wk[t].weak_blocks = weak_blocks;
        wk[t].strong_first = (size_t)t * base + ((size_t)t < rem ? (size_t)t : rem);
        wk[t].strong_blocks = base + ((size_t)t < rem ? 1 : 0);
        if (pthread_create(&th[t], NULL, worker, &wk[t]) != 0) { fprintf(stderr, "pthread_create failed\n"); return 2; }
    }
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);

    int bad = 0;
    double weak_wall = 0, strong_wall = 0, weak_bytes = 0, strong_bytes = 0;
    for (int t = 0; t < n_threads; ++t) {
        if (wk[t].status) { bad = wk[t].status; fprintf(stderr, "thread %d (device %d): %s\n", t, wk[t].device, wk[t].err); }
        if (wk[t].weak_wall_s > weak_wall) weak_wall = wk[t].weak_wall_s;
        if (wk[t].strong_wall_s > strong_wall) strong_wall = wk[t].strong_wall_s;
        weak_bytes += wk[t].weak_bytes; strong_bytes += wk[t].strong_bytes;
    }
    printf("{\"devices\": %d, \"threads\": %d, \"steps\": %d, \"correct\": %s, \"collective\": \"none\",", n_dev, n_threads, steps, bad ? "false" : "true");
    printf(" \"weak_u32_w7_unpack\": {\"blocks_per_thread\": %zu, \"Gint_per_s\": %.2f, \"aggregate_GBps\": %.1f, \"per_thread\": [",
           weak_blocks, weak_wall > 0 ? (double)weak_blocks * 1024.0 * n_threads * steps / weak_wall / 1e9 : 0.0,
           weak_wall > 0 ? weak_bytes * steps / weak_wall / 1e9 : 0.0);
    for (int t = 0; t < n_threads; ++t)
        printf("%s{\"thread\": %d, \"device\": %d, \"kernel_ms\": %.4f, \"GBps\": %.1f, \"correct\": %s}", t ? ", " : "", t, wk[t].device,
               wk[t].weak_kernel_ms, wk[t].weak_kernel_ms > 0 ? wk[t].weak_bytes / wk[t].weak_kernel_ms / 1e6 : 0.0, wk[t].status ? "false" : "true");
    printf("]}, \"strong_u32_mixed_unpack\": {\"blocks_total\": %zu, \"Gint_per_s\": %.2f, \"aggregate_GBps\": %.1f, \"per_thread\": [",
           strong_total, strong_wall > 0 ? (double)strong_total * 1024.0 * steps / strong_wall / 1e9 : 0.0,
           strong_wall > 0 ? strong_bytes * steps / strong_wall / 1e9 : 0.0);
    for (int t = 0; t < n_threads; ++t)
        printf("%s{\"thread\": %d, \"device\": %d, \"first_block\": %zu, \"blocks\": %zu, \"kernel_ms\": %.4f, \"GBps\": %.1f, \"correct\": %s}",
               t ? ", " : "", t, wk[t].device, wk[t].strong_first, wk[t].strong_blocks, wk[t].strong_kernel_ms,
               wk[t].strong_kernel_ms > 0 ? wk[t].strong_bytes / wk[t].strong_kernel_ms / 1e6 : 0.0, wk[t].status ? "false" : "true");
    printf("]}}\n");
    pthread_barrier_destroy(&bar);
    return bad;
}
