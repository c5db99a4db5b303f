/*
 * selftest.c -- sanitizer run of the CPU oracle (test infrastructure).
 * Built with -fsanitize=address,undefined by tests/test_oracle_sanitizers.py: exercises every
 * literal and fast entry point for all (T,W) on random full-width data, so that shifts by >= the
 * type width, misaligned or out-of-bounds accesses and signed overflow in the restatement would
 * abort.  Also re-checks the round-trip / closed-form invariants natively.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fl_oracle.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd(void)
{
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static int failures = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

#define SELFTEST(T, S, TB)                                                                                   \
    static void selftest_##S(void)                                                                           \
    {                                                                                                        \
        enum { L = 1024 / TB, NB = 3 };                                                                      \
        T *v = malloc(NB * 1024 * sizeof(T)), *pk = malloc(NB * 1024 * sizeof(T)), *un = malloc(NB * 1024 * sizeof(T)); \
        T *t2 = malloc(NB * 1024 * sizeof(T)), *base = malloc(NB * L * sizeof(T)), *refs = malloc(NB * sizeof(T));       \
        for (unsigned w = 0; w <= TB; ++w) {                                                                 \
            for (int i = 0; i < NB * 1024; ++i) v[i] = (T)rnd();                                             \
            for (int i = 0; i < NB * L; ++i) base[i] = (T)rnd();                                             \
            for (int i = 0; i < NB; ++i) refs[i] = (T)rnd();                                                 \
            const size_t pl = 1024u * w / TB;                                                                \
            const T mask = w == TB ? (T) ~(T)0 : (T)(((T)1 << w) - 1);                                       \
            CHECK(fl_oracle_pack_##S(w, v, pk) == 0);                                                        \
            CHECK(fl_oracle_unpack_##S(w, pk, un) == 0);                                                     \
            for (int i = 0; i < 1024; ++i) {                                                                 \
                T one;                                                                                       \
                CHECK(un[i] == (T)(v[i] & mask));                                                            \
                CHECK(fl_oracle_unpack_single_##S(w, pk, (size_t)i, &one) == 0 && one == un[i]);             \
            }                                                                                                \
            CHECK(fl_oracle_for_pack_##S(w, v, refs[0], t2) == 0);                                           \
            CHECK(fl_oracle_unfor_pack_##S(w, t2, refs[0], un) == 0);                                        \
            for (int i = 0; i < 1024; ++i) CHECK(un[i] == (T)((T)((T)(v[i] - refs[0]) & mask) + refs[0]));   \
            CHECK(fl_oracle_undelta_pack_##S(w, pk, base, un) == 0);                                         \
            CHECK(fl_oracle_unpack_##S(w, pk, t2) == 0 && fl_oracle_undelta_##S(t2, base, t2 + 1024) == 0);  \
            CHECK(memcmp(un, t2 + 1024, 1024 * sizeof(T)) == 0);                                             \
            /* fast family: specialised pairs must equal the literal one; others report 4 */                 \
            int rc = fl_oracle_fast_pack_##S(w, v, t2, NB, 2);                                               \
            if (rc == 0) {                                                                                   \
                for (int b = 0; b < NB; ++b) {                                                               \
                    CHECK(fl_oracle_pack_##S(w, v + b * 1024, pk) == 0);                                     \
                    CHECK(memcmp(pk, t2 + b * pl, pl * sizeof(T)) == 0);                                     \
                }                                                                                            \
            } else CHECK(rc == 4);                                                                           \
            for (size_t i = 0; i < NB * pl; ++i) pk[i] = (T)rnd();                                           \
            rc = fl_oracle_fast_unpack_##S(w, pk, t2, NB, 2);                                                \
            if (rc == 0) {                                                                                   \
                for (int b = 0; b < NB; ++b) {                                                               \
                    CHECK(fl_oracle_unpack_##S(w, pk + b * pl, un) == 0);                                    \
                    CHECK(memcmp(un, t2 + b * 1024, 1024 * sizeof(T)) == 0);                                 \
                }                                                                                            \
            } else CHECK(rc == 4);                                                                           \
            rc = fl_oracle_fast_undelta_pack_##S(w, pk, base, t2, NB, 2);                                    \
            CHECK(rc == 0 || rc == 4);                                                                       \
            rc = fl_oracle_fast_unfor_pack_##S(w, pk, refs, t2, NB, 2);                                      \
            CHECK(rc == 0 || rc == 4);                                                                       \
        }                                                                                                    \
        CHECK(fl_oracle_pack_##S(TB + 1, v, pk) == 1);                                                       \
        CHECK(fl_oracle_delta_##S(v, base, un) == 0 && fl_oracle_undelta_##S(un, base, t2) == 0);            \
        CHECK(memcmp(v, t2, 1024 * sizeof(T)) == 0);                                                         \
        CHECK(fl_oracle_transpose_##S(v, un) == 0 && fl_oracle_untranspose_##S(un, t2) == 0);                \
        CHECK(memcmp(v, t2, 1024 * sizeof(T)) == 0);                                                         \
        free(v); free(pk); free(un); free(t2); free(base); free(refs);                                       \
    }

SELFTEST(uint8_t, u8, 8)
SELFTEST(uint16_t, u16, 16)
SELFTEST(uint32_t, u32, 32)
SELFTEST(uint64_t, u64, 64)

int main(void)
{
    selftest_u8();
    selftest_u16();
    selftest_u32();
    selftest_u64();
    uint64_t buf[4 * 16];
    CHECK(fl_oracle_parallel_fill(buf, 128, 4, 1, 3) == 0);
    printf(failures ? "FAILED (%d)\n" : "ok\n", failures);
    return failures ? 1 : 0;
}
