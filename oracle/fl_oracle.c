/*
 * fl_oracle.c -- CPU oracle for the FastLanes 1024-value codec hot path.
 * TEST INFRASTRUCTURE ONLY; see fl_oracle.h for scope and parity status.
 * Restates /root/reference/src/{lib,macros,bitpacking,delta,ffor,transpose}.rs
 * (each function cites the lines it follows).  No reference source is copied:
 * the Rust macros are re-expressed as plain C loops.
 */
#include "fl_oracle.h"
#include <pthread.h>

/* lib.rs:22 */
const unsigned fl_oracle_FL_ORDER[8] = {0, 4, 2, 6, 1, 5, 3, 7};

/* macros.rs:20-24 (duplicated at :46-50 and :112-116) */
unsigned fl_oracle_index(unsigned row, unsigned lane)
{
    unsigned o = row / 8;
    unsigned s = row % 8;
    return fl_oracle_FL_ORDER[o] * 16 + s * 128 + lane;
}

/* transpose.rs:29-36 */
unsigned fl_oracle_transpose_index(unsigned idx)
{
    unsigned lane = idx % 16;
    unsigned order = (idx / 16) % 8;
    unsigned row = idx / 128;
    return lane * 64 + fl_oracle_FL_ORDER[order] * 8 + row;
}

typedef struct { uint64_t *p; size_t w0, w1; uint64_t seed; } fill_job_t;

static void *fill_worker(void *arg)
{
    fill_job_t *j = (fill_job_t *)arg;
    for (size_t i = j->w0; i < j->w1; ++i) {
        uint64_t z = j->seed + (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        j->p[i] = z ^ (z >> 31);
    }
    return NULL;
}

int fl_oracle_parallel_fill(void *base, size_t bytes_per_block, size_t n_blocks, uint64_t seed,
                            unsigned nthreads)
{
    if (bytes_per_block % 8) return 1;
    if (nthreads == 0) nthreads = 1;
    if (nthreads > 1024) nthreads = 1024;
    if (nthreads > n_blocks) nthreads = n_blocks ? (unsigned)n_blocks : 1;
    fill_job_t jobs[1024];
    pthread_t tids[1024];
    const size_t wpb = bytes_per_block / 8;
    for (unsigned t = 0; t < nthreads; ++t) {
        jobs[t].p = (uint64_t *)base;
        jobs[t].w0 = (n_blocks * t / nthreads) * wpb;
        jobs[t].w1 = (n_blocks * (t + 1) / nthreads) * wpb;
        jobs[t].seed = seed;
    }
    if (nthreads == 1) { fill_worker(&jobs[0]); return 0; }
    for (unsigned t = 0; t < nthreads; ++t)
        if (pthread_create(&tids[t], NULL, fill_worker, &jobs[t]) != 0) return 3;
    for (unsigned t = 0; t < nthreads; ++t) pthread_join(tids[t], NULL);
    return 0;
}

/* What the spliced unpack! body does with each (idx, elem). */
enum { FL_BODY_STORE = 0, FL_BODY_ADD_REF = 1, FL_BODY_UNDELTA = 2 };

#define T_ uint8_t
#define S_ u8
#define TB_ 8u
#define LN_ 128u
#define FL_FAST_LIST_ FL_FAST_(0, 3) FL_FAST_(1, 3)
#include "fl_oracle_impl.inc"
#undef T_
#undef S_
#undef TB_
#undef LN_
#undef FL_FAST_LIST_

#define T_ uint16_t
#define S_ u16
#define TB_ 16u
#define LN_ 64u
/* benches/bitpacking.rs (W=3), benches/delta.rs (W=9 fused undelta) */
#define FL_FAST_LIST_ FL_FAST_(0, 3) FL_FAST_(1, 3) FL_FAST_(3, 9)
#include "fl_oracle_impl.inc"
#undef T_
#undef S_
#undef TB_
#undef LN_
#undef FL_FAST_LIST_

#define T_ uint32_t
#define S_ u32
#define TB_ 32u
#define LN_ 32u
/* BASELINE.json configs 2 and 4 (+ W=10 of bitpacking.rs:248-256) */
#define FL_FAST_LIST_ FL_FAST_(0, 7) FL_FAST_(1, 7) FL_FAST_(2, 7) FL_FAST_(1, 10) \
    FL_FAST_(0, 12) FL_FAST_(1, 12) FL_FAST_(3, 12)
/* BASELINE.json config 5 (u32, every width): fl_oracle_fast_unpack_mixed_u32 */
#define FL_MIXED_CASES_ \
    FL_MIXED_(0) FL_MIXED_(1) FL_MIXED_(2) FL_MIXED_(3) FL_MIXED_(4) FL_MIXED_(5) FL_MIXED_(6) FL_MIXED_(7)       \
    FL_MIXED_(8) FL_MIXED_(9) FL_MIXED_(10) FL_MIXED_(11) FL_MIXED_(12) FL_MIXED_(13) FL_MIXED_(14) FL_MIXED_(15)  \
    FL_MIXED_(16) FL_MIXED_(17) FL_MIXED_(18) FL_MIXED_(19) FL_MIXED_(20) FL_MIXED_(21) FL_MIXED_(22) FL_MIXED_(23) \
    FL_MIXED_(24) FL_MIXED_(25) FL_MIXED_(26) FL_MIXED_(27) FL_MIXED_(28) FL_MIXED_(29) FL_MIXED_(30) FL_MIXED_(31) \
    FL_MIXED_(32)
#include "fl_oracle_impl.inc"
#undef FL_MIXED_CASES_
#undef T_
#undef S_
#undef TB_
#undef LN_
#undef FL_FAST_LIST_

#define T_ uint64_t
#define S_ u64
#define TB_ 64u
#define LN_ 16u
/* BASELINE.json config 3 */
#define FL_FAST_LIST_ FL_FAST_(0, 17) FL_FAST_(1, 17)
#include "fl_oracle_impl.inc"
