/* column_decode.c -- the drop-in boundary used from plain C (what a cgo / JNI / Rust-FFI binding would call):
 * a caller that today loops the reference's trait methods over 1024-value chunks,
 *
 *     for b in 0..n { u32::unchecked_unpack(width[b], &packed[off[b]..], &mut out[b*1024..]) }     (bitpacking.rs:109-129)
 *     let v = u32::unchecked_unpack_single(width[b], &packed[off[b]..], i);                        (bitpacking.rs:181-200)
 *     for b in 0..n { u32::for_pack::<W_b>(&v[b*1024..], min_b, ..) / unfor_pack::<W_b>(..) }                (ffor.rs:24-50)
 *
 * keeps its column in HBM and makes ONE call per loop.  Everything below is the C ABI of include/fastlanes_amd.h plus
 * the HIP runtime for memory; no C++.
 *
 * Build: gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/column_decode.c \
 *            -L fastlanes_amd -lfastlanes_amd -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../fastlanes_amd' -o examples/column_decode
 * Run (needs a GPU): prints "ok".
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "fastlanes_amd.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); return 2; } } while (0)
#define CHECK_FL(x) do { int s_ = (x); if (s_ != FL_OK) { printf("%s at line %d\n", fl_status_string(s_), __LINE__); return 2; } } while (0)

int main(void)
{
    enum { N = 1000 };                                   /* blocks: 1 024 000 values */
    /* the values to encode: block b holds numbers below 2^width[b]; widths as a columnar writer would pick them */
    uint8_t *widths = malloc(N);
    uint32_t *values = malloc((size_t)N * 1024 * 4);
    for (int b = 0; b < N; ++b) {
        widths[b] = (uint8_t)(1 + (b * 7) % 32);
        const uint32_t mask = widths[b] == 32 ? 0xFFFFFFFFu : ((1u << widths[b]) - 1u);
        for (int i = 0; i < 1024; ++i) values[(size_t)b * 1024 + i] = (uint32_t)((b * 1024u + i) * 2654435761u) & mask;
    }
    /* device-resident column: widths, offsets (built on the device), packed bytes, decoded values */
    uint8_t *d_widths; uint64_t *d_offsets, *d_total; uint32_t *d_err, *d_values, *d_packed, *d_decoded, *d_picked; uint64_t *d_idx;
    CHECK_HIP(hipMalloc((void **)&d_widths, N));
    CHECK_HIP(hipMalloc((void **)&d_offsets, N * sizeof(uint64_t)));
    CHECK_HIP(hipMalloc((void **)&d_total, sizeof(uint64_t)));
    CHECK_HIP(hipMalloc((void **)&d_err, sizeof(uint32_t)));
    CHECK_HIP(hipMemset(d_err, 0, sizeof(uint32_t)));
    CHECK_HIP(hipMalloc((void **)&d_values, (size_t)N * 4096));
    CHECK_HIP(hipMalloc((void **)&d_decoded, (size_t)N * 4096));
    CHECK_HIP(hipMemcpy(d_widths, widths, N, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_values, values, (size_t)N * 4096, hipMemcpyHostToDevice));
    CHECK_FL(fl_widths_to_offsets(32, d_widths, N, d_offsets, d_total, d_err, NULL));
    uint64_t total = 0;
    CHECK_HIP(hipMemcpy(&total, d_total, sizeof total, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMalloc((void **)&d_packed, total));
    /* encode: the unchecked_pack loop (bitpacking.rs:76-96); decode: the unchecked_unpack loop; both one call */
    CHECK_FL(fl_u32_pack_widths(d_widths, d_offsets, d_values, d_packed, total, N, d_err, NULL));
    CHECK_FL(fl_u32_unpack_widths(d_widths, d_offsets, d_packed, total, d_decoded, N, d_err, NULL));
    /* point lookups: unchecked_unpack_single for a few global element indices */
    const uint64_t idx[4] = {0, 1023, 517 * 1024 + 77, (uint64_t)N * 1024 - 1};
    CHECK_HIP(hipMalloc((void **)&d_idx, sizeof idx));
    CHECK_HIP(hipMalloc((void **)&d_picked, 4 * sizeof(uint32_t)));
    CHECK_HIP(hipMemcpy(d_idx, idx, sizeof idx, hipMemcpyHostToDevice));
    CHECK_FL(fl_u32_unpack_single_widths(d_widths, d_offsets, d_packed, total, N, d_idx, 4, d_picked, d_err, NULL));
    /* a FoR column, widths chosen ON THE DEVICE: the same values shifted by a per-block frame of reference; the encoder finds every
     * block's minimum and maximum, takes the bit length of their difference as the width, lays the blocks out back to back and packs
     * `value - minimum` (for_pack::<W>, ffor.rs:24-36); the decoder adds the minimum back (unfor_pack::<W>, :38-50).  Five calls, no
     * host pass over the column. */
    uint32_t *shifted = malloc((size_t)N * 4096);
    for (int b = 0; b < N; ++b)
        for (int i = 0; i < 1024; ++i) shifted[(size_t)b * 1024 + i] = values[(size_t)b * 1024 + i] / 2u + 1000003u * (uint32_t)b;
    uint32_t *d_shifted, *d_mins, *d_maxs, *d_for_packed, *d_for_decoded; uint8_t *d_for_widths; uint64_t *d_for_offsets;
    CHECK_HIP(hipMalloc((void **)&d_shifted, (size_t)N * 4096));
    CHECK_HIP(hipMalloc((void **)&d_for_decoded, (size_t)N * 4096));
    CHECK_HIP(hipMalloc((void **)&d_mins, N * sizeof(uint32_t)));
    CHECK_HIP(hipMalloc((void **)&d_maxs, N * sizeof(uint32_t)));
    CHECK_HIP(hipMalloc((void **)&d_for_widths, N));
    CHECK_HIP(hipMalloc((void **)&d_for_offsets, N * sizeof(uint64_t)));
    CHECK_HIP(hipMemcpy(d_shifted, shifted, (size_t)N * 4096, hipMemcpyHostToDevice));
    CHECK_FL(fl_u32_block_min_max(d_shifted, N, d_mins, d_maxs, NULL));
    CHECK_FL(fl_u32_for_widths(d_mins, d_maxs, N, d_for_widths, NULL));
    CHECK_FL(fl_widths_to_offsets(32, d_for_widths, N, d_for_offsets, d_total, d_err, NULL));
    uint64_t for_total = 0;
    CHECK_HIP(hipMemcpy(&for_total, d_total, sizeof for_total, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMalloc((void **)&d_for_packed, for_total ? for_total : 16));
    CHECK_FL(fl_u32_for_pack_widths(d_for_widths, d_for_offsets, d_shifted, d_mins, 1, d_for_packed, for_total, N, d_err, NULL));
    CHECK_FL(fl_u32_unfor_pack_widths(d_for_widths, d_for_offsets, d_for_packed, for_total, d_mins, 1, d_for_decoded, N, d_err, NULL));
    uint32_t *for_decoded = malloc((size_t)N * 4096);
    CHECK_HIP(hipMemcpy(for_decoded, d_for_decoded, (size_t)N * 4096, hipMemcpyDeviceToHost));
    const int for_bad = memcmp(for_decoded, shifted, (size_t)N * 4096) != 0 || for_total >= total;   /* halved values: smaller than the plain column */
    /* the same trait call on host slices, one block (what a trait-for-trait binding does): block 517 */
    uint64_t off517 = 0;
    CHECK_HIP(hipMemcpy(&off517, d_offsets + 517, sizeof off517, hipMemcpyDeviceToHost));
    const unsigned w517 = widths[517];
    uint32_t *one_packed = malloc(128u * w517), one_block[1024];
    CHECK_HIP(hipMemcpy(one_packed, (char *)d_packed + off517, 128u * w517, hipMemcpyDeviceToHost));
    CHECK_FL(fl_u32_unpack_host(w517, one_packed, one_block, 1));
    /* check everything against the values we started from */
    uint32_t *decoded = malloc((size_t)N * 4096), picked[4], err = 0;
    CHECK_HIP(hipMemcpy(decoded, d_decoded, (size_t)N * 4096, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(picked, d_picked, sizeof picked, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(&err, d_err, sizeof err, hipMemcpyDeviceToHost));
    int bad = err != 0 || for_bad || memcmp(decoded, values, (size_t)N * 4096) != 0 || memcmp(one_block, values + 517 * 1024, 4096) != 0;
    for (int k = 0; k < 4; ++k) bad |= picked[k] != values[idx[k]];
    printf(bad ? "MISMATCH\n" : "ok\n");
    fl_host_release();
    return bad;
}
