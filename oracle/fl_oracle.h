/*
 * fl_oracle.h -- CPU oracle for the FastLanes 1024-value codec hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * algorithm (spiraldb/fastlanes v0.1.8, /root/reference/src).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it; the
 * product library (fastlanes_amd/csrc) never links or loads it.
 *
 * PARITY STATUS: the reference is a Rust crate (nightly-2024-06-19 +
 * generic_const_exprs + un-vendored crates.io deps) and no Rust toolchain is
 * present, so no reference-executed bytes exist here.  The reference also
 * ships no golden byte vectors.  The oracle is pinned by
 *   (1) every assertion of every reference unit test, re-run against it
 *       (tests/test_oracle_reference_tests.py),
 *   (2) the closed-form reader `unpack_single` (bitpacking.rs:132-179), an
 *       independent spec of the wire format, inverting pack for all 124
 *       (T,W) pairs on random full-width data,
 *   (3) an independent bit-level numpy model (tests/bitmodel.py), and
 *   (4) the model-derived known-answer vectors of SURVEY.md section 8(c).
 * "Pinned by reference-executed output" it is NOT; DESIGN.md says the same.
 *
 * Two families are exported per element type u8/u16/u32/u64:
 *   fl_oracle_<op>_<ty>      literal restatement (lane-outer / row-inner,
 *                            runtime width) -- THE checker.
 *   fl_oracle_fast_<op>_<ty> the same arithmetic with the width constant-
 *                            folded and the lane loop innermost so gcc
 *                            auto-vectorises it, as LLVM does for the
 *                            reference (README.md:9-10) -- used only as the
 *                            timed CPU baseline ("port").
 * All functions return 0, or 1 when width > T (reference: unreachable!()
 * panic, bitpacking.rs:93,126,197), or 2 when index >= 1024
 * (bitpacking.rs:152 assert!).
 */
#ifndef FL_ORACLE_H
#define FL_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* lib.rs:22 */
extern const unsigned fl_oracle_FL_ORDER[8];
/* macros.rs:20-24 */
unsigned fl_oracle_index(unsigned row, unsigned lane);
/* transpose.rs:29-36 */
unsigned fl_oracle_transpose_index(unsigned idx);

/* CPU-baseline helper: fill n_blocks * bytes_per_block bytes at base with splitmix64 output,
 * using the SAME contiguous block-range-per-thread partition as the fl_oracle_fast_* functions,
 * so that every page is first touched (NUMA-placed) by the thread that will stream it. */
int fl_oracle_parallel_fill(void *base, size_t bytes_per_block, size_t n_blocks, uint64_t seed,
                            unsigned nthreads);

#define FL_ORACLE_DECL(T, S)                                                              \
    int fl_oracle_pack_##S(unsigned width, const T *in, T *out);                          \
    int fl_oracle_unpack_##S(unsigned width, const T *in, T *out);                        \
    int fl_oracle_unpack_single_##S(unsigned width, const T *packed, size_t index,        \
                                    T *value);                                            \
    int fl_oracle_for_pack_##S(unsigned width, const T *in, T reference, T *out);         \
    int fl_oracle_unfor_pack_##S(unsigned width, const T *in, T reference, T *out);       \
    int fl_oracle_delta_##S(const T *in, const T *base, T *out);                          \
    int fl_oracle_undelta_##S(const T *in, const T *base, T *out);                        \
    int fl_oracle_undelta_pack_##S(unsigned width, const T *in, const T *base, T *out);   \
    int fl_oracle_transpose_##S(const T *in, T *out);                                     \
    int fl_oracle_untranspose_##S(const T *in, T *out);                                   \
    /* batched helpers: n_blocks contiguous blocks, nthreads pthreads */                  \
    int fl_oracle_fast_pack_##S(unsigned width, const T *in, T *out, size_t n_blocks,     \
                                unsigned nthreads);                                       \
    int fl_oracle_fast_unpack_##S(unsigned width, const T *in, T *out, size_t n_blocks,   \
                                  unsigned nthreads);                                     \
    int fl_oracle_fast_for_pack_##S(unsigned width, const T *in, const T *refs,           \
                                    T *out, size_t n_blocks, unsigned nthreads);          \
    int fl_oracle_fast_unfor_pack_##S(unsigned width, const T *in, const T *refs,         \
                                      T *out, size_t n_blocks, unsigned nthreads);        \
    int fl_oracle_fast_undelta_pack_##S(unsigned width, const T *in, const T *bases,      \
                                        T *out, size_t n_blocks, unsigned nthreads);      \
    /* per-block content hashes of an unpacked column (full-size GPU-vs-oracle check):     \
     * sum[b] = sum_i v[i], wsum[b] = sum_i (i+1)*v[i], both wrapping uint64 */            \
    int fl_oracle_block_hashes_##S(const T *v, size_t n_blocks, uint64_t *sum,            \
                                   uint64_t *wsum, unsigned nthreads);

/* CPU-baseline helper for BASELINE.json config 5 (u32 only): the reference's caller loop over
 * per-block widths (bitpacking.rs:109-129), every width 0..=32 specialised like the fast family;
 * offsets are byte offsets into `packed`; returns 1 on a width > 32. */
int fl_oracle_fast_unpack_mixed_u32(const uint8_t *widths, const uint64_t *offsets,
                                    const uint32_t *packed, uint32_t *out, size_t n_blocks,
                                    unsigned nthreads);

FL_ORACLE_DECL(uint8_t, u8)
FL_ORACLE_DECL(uint16_t, u16)
FL_ORACLE_DECL(uint32_t, u32)
FL_ORACLE_DECL(uint64_t, u64)

#ifdef __cplusplus
}
#endif
#endif
