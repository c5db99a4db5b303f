// The reference's own unit tests, written against the C++ trait
// mirror (include/fastlanes_amd.hpp) so they read like the Rust originals.  Needs a GPU
// at run time (no CPU path exists); built on CPU as a compile/link check.
//   g++ -std=c++17 -I include -I /opt/rocm/include tests/cpp/test_trait_mirror.cpp -L fastlanes_amd -lfastlanes_amd -L /opt/rocm/lib -lamdhip64
#include <cstdio>
#include <cstring>
#include <utility>
#include <vector>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "fastlanes_amd.hpp"

using namespace fastlanes;

static int failures = 0;
static int round_trips = 0;
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

// lib.rs:53-59
static void test_ordering_is_own_inverse() { for (int i = 0; i < 8; ++i) EXPECT(FL_ORDER[FL_ORDER[i]] == (size_t)i); }

// lib.rs:71-96 / README.md:14-47
static void pack_u16_into_u3_no_unsafe()
{
    constexpr size_t WIDTH = 3;
    static uint16_t values[1024], packed[128 * WIDTH / sizeof(uint16_t)], unpacked[1024];
    for (int i = 0; i < 1024; ++i) values[i] = (uint16_t)(i % (1 << WIDTH));
    BitPacking<uint16_t>::pack<WIDTH>(values, packed);
    BitPacking<uint16_t>::unpack<WIDTH>(packed, unpacked);
    EXPECT(std::memcmp(values, unpacked, sizeof values) == 0);
    for (int i = 0; i < 1024; i += 37) EXPECT(BitPacking<uint16_t>::unpack_single<WIDTH>(packed, i) == values[i]);
    // the unchecked_ forms of the README
    std::vector<uint16_t> p2(192), u2(1024);
    BitPacking<uint16_t>::unchecked_pack(WIDTH, values, 1024, p2.data(), p2.size());
    BitPacking<uint16_t>::unchecked_unpack(WIDTH, p2.data(), p2.size(), u2.data(), u2.size());
    EXPECT(std::memcmp(values, u2.data(), sizeof values) == 0);
    EXPECT(BitPacking<uint16_t>::unchecked_unpack_single(WIDTH, p2.data(), p2.size(), 14) == values[14]);
    // KAT-2
    EXPECT(packed[0] == 0x0000 && packed[1] == 0x9249 && packed[2] == 0x2492 && packed[3] == 0xB6DB);
}

// bitpacking.rs:248-256
static void test_unchecked_pack()
{
    std::vector<uint32_t> input(1024), packed(320), output(1024);
    for (int i = 0; i < 1024; ++i) input[i] = i;
    BitPacking<uint32_t>::unchecked_pack(10, input.data(), 1024, packed.data(), 320);
    BitPacking<uint32_t>::unchecked_unpack(10, packed.data(), 320, output.data(), 1024);
    EXPECT(input == output);
    EXPECT(packed[0] == 0x10020000u && packed[319] == 0xFFF7FBFEu);   // KAT-3
}

// bitpacking.rs:258-271
static void test_unpack_single()
{
    static uint32_t values[1024], packed[512];
    for (int i = 0; i < 1024; ++i) values[i] = i;
    BitPacking<uint32_t>::pack<16>(values, packed);
    for (int i = 0; i < 1024; ++i) {        // every index, as bitpacking.rs:264
        EXPECT(BitPacking<uint32_t>::unpack_single<16>(packed, i) == values[i]);
        EXPECT(BitPacking<uint32_t>::unchecked_unpack_single(16, packed, 512, i) == values[i]);
    }
}

// bitpacking.rs:273-315: try_round_trip::<T, W>() for every (T, W) -- the reference's 124-case matrix
// (seq!(W in 0..=8) for u8 ... seq!(W in 0..=64) for u64), unpack_single for EVERY index, through the trait mirror.
template <typename T, size_t W> static void try_round_trip()
{
    constexpr size_t TB = sizeof(T) * 8;
    static T values[1024], packed[W ? 1024 * W / TB : 1], unpacked[1024];
    for (size_t i = 0; i < 1024; ++i) values[i] = (T)(i % ((size_t)1 << (W % TB)));   // bitpacking.rs:281
    BitPacking<T>::template pack<W>(values, packed);
    std::memset(unpacked, 0xEE, sizeof unpacked);
    BitPacking<T>::template unpack<W>(packed, unpacked);
    EXPECT(std::memcmp(unpacked, values, sizeof values) == 0);
    int bad = 0;
    for (size_t i = 0; i < 1024; ++i) {
        bad += BitPacking<T>::template unpack_single<W>(packed, i) != values[i];
        bad += BitPacking<T>::unchecked_unpack_single(W, packed, 1024 * W / TB, i) != values[i];
    }
    if (bad) { std::printf("FAIL try_round_trip<u%zu, %zu>: %d unpack_single mismatches\n", TB, W, bad); ++failures; }
    ++round_trips;
}
template <typename T, size_t... Ws> static void round_trip_all(std::index_sequence<Ws...>) { (try_round_trip<T, Ws>(), ...); }
static void test_round_trip_matrix()
{
    round_trip_all<uint8_t>(std::make_index_sequence<9>{});
    round_trip_all<uint16_t>(std::make_index_sequence<17>{});
    round_trip_all<uint32_t>(std::make_index_sequence<33>{});
    round_trip_all<uint64_t>(std::make_index_sequence<65>{});
    EXPECT(round_trips == 124);
}

// delta.rs:80-107
static void test_delta()
{
    constexpr size_t W = 15;
    static uint16_t values[1024], transposed[1024], deltas[1024], packed[128 * W / 2], unpacked[1024], undelta[1024];
    static uint16_t zero[64] = {0};
    for (int i = 0; i < 1024; ++i) values[i] = (uint16_t)(i / 8);
    Transpose<uint16_t>::transpose(values, transposed);
    Delta<uint16_t>::delta(transposed, zero, deltas);
    BitPacking<uint16_t>::pack<W>(deltas, packed);
    Delta<uint16_t>::undelta_pack<W>(packed, zero, unpacked);          // fused kernel
    EXPECT(std::memcmp(transposed, unpacked, sizeof unpacked) == 0);
    BitPacking<uint16_t>::unpack<W>(packed, unpacked);                 // unfused
    Delta<uint16_t>::undelta(unpacked, zero, undelta);
    EXPECT(std::memcmp(transposed, undelta, sizeof undelta) == 0);
    static uint16_t back[1024];
    Transpose<uint16_t>::untranspose(transposed, back);
    EXPECT(std::memcmp(values, back, sizeof back) == 0);
    for (int i = 0; i < 1024; i += 101) EXPECT(transposed[i] == values[transpose(i)]);
}

// ffor.rs:66-88
static void test_ffor()
{
    constexpr size_t W = 15;
    static uint16_t values[1024], packed[128 * W / 2], unpacked[1024];
    for (int i = 0; i < 1024; ++i) values[i] = (uint16_t)(i % (1 << W));
    FoR<uint16_t>::for_pack<W>(values, 10, packed);
    BitPacking<uint16_t>::unpack<W>(packed, unpacked);
    for (int i = 0; i < 1024; ++i) EXPECT((uint16_t)((values[i] - 10) & ((1 << W) - 1)) == unpacked[i]);
    FoR<uint16_t>::unfor_pack<W>(packed, 10, unpacked);
    for (int i = 0; i < 1024; ++i) EXPECT((uint16_t)(((values[i] - 10) & 0x7FFF) + 10) == unpacked[i]);
}

// the reference panics (bitpacking.rs:93 unreachable!, :152 assert!)
static void test_panics()
{
    std::vector<uint32_t> in(1024), out(1024);
    bool threw = false;
    try { BitPacking<uint32_t>::unchecked_pack(33, in.data(), 1024, out.data(), 1024); } catch (const Error& e) { threw = e.status == FL_ERR_WIDTH; }
    EXPECT(threw);
    threw = false;
    static uint32_t packed[96];
    try { (void)BitPacking<uint32_t>::unpack_single<3>(packed, 1024); } catch (const Error& e) { threw = e.status == FL_ERR_INDEX; }
    EXPECT(threw);
}

// u64 W=17 KAT-5 and u8
static void test_wide_and_narrow()
{
    static uint64_t v[1024], pk[272], un[1024];
    for (uint64_t i = 0; i < 1024; ++i) v[i] = (i * 2654435761ull) & 0x1FFFF;
    BitPacking<uint64_t>::pack<17>(v, pk);
    EXPECT(pk[0] == 0x4C06C401B1000000ull && pk[1] == 0x198CAAC4A46379B1ull);
    BitPacking<uint64_t>::unpack<17>(pk, un);
    EXPECT(std::memcmp(v, un, sizeof v) == 0);
    static uint8_t v8[1024], p8[1024], u8_[1024];
    for (int i = 0; i < 1024; ++i) v8[i] = (uint8_t)i;
    BitPacking<uint8_t>::pack<8>(v8, p8);
    EXPECT(std::memcmp(v8, p8, 1024) == 0);   // KAT-6: the copy path is the identity for u8
    BitPacking<uint8_t>::unpack<8>(p8, u8_);
    EXPECT(std::memcmp(v8, u8_, 1024) == 0);
}

// Device tier + extensions through the C++ mirror: the batched form of benches/bitpacking.rs:80-97,
// fused Delta decode to original order, mixed widths.
template <typename U> struct DevVec {
    U* p = nullptr; size_t n;
    explicit DevVec(size_t n_) : n(n_) { if (hipMalloc((void**)&p, n * sizeof(U)) != hipSuccess) throw std::runtime_error("hipMalloc"); }
    ~DevVec() { (void)hipFree(p); }
    void up(const std::vector<U>& h) { (void)hipMemcpy(p, h.data(), n * sizeof(U), hipMemcpyHostToDevice); }
    std::vector<U> down() const { std::vector<U> h(n); (void)hipMemcpy(h.data(), p, n * sizeof(U), hipMemcpyDeviceToHost); return h; }
};

static void test_device_tier()
{
    const size_t N = 100;   // blocks
    std::vector<uint32_t> v(N * 1024);
    for (size_t i = 0; i < v.size(); ++i) v[i] = (uint32_t)((i * 2654435761ull) >> 7) & 0x7F;
    DevVec<uint32_t> dv(N * 1024), dp(N * 224), du(N * 1024);
    dv.up(v);
    BitPacking<uint32_t>::pack_device(7, dv.p, dp.p, N);
    BitPacking<uint32_t>::unpack_device(7, dp.p, du.p, N);
    EXPECT(du.down() == v);
    // per-block sums of the decoded values == sums of the originals
    DevVec<uint64_t> ds(N);
    BitPacking<uint32_t>::unpack_block_sums_device(7, dp.p, N, ds.p);
    auto sums = ds.down();
    for (size_t b = 0; b < N; ++b) { uint64_t s = 0; for (int i = 0; i < 1024; ++i) s += v[b * 1024 + i]; EXPECT(sums[b] == s); }
    // selection mask straight from packed data: v < 64
    DevVec<uint32_t> dmask(N * 32);
    BitPacking<uint32_t>::unpack_compare_device(7, dp.p, FL_CMP_LT, 64u, N, dmask.p);
    auto mask = dmask.down();
    for (size_t i = 0; i < v.size(); i += 97) EXPECT(((mask[i / 32] >> (i % 32)) & 1u) == (v[i] < 64u ? 1u : 0u));
    // fused encode/decode in the ORIGINAL order round-trips (W = 32: lossless for any data)
    std::vector<uint32_t> bases(N * 32, 12345u);
    DevVec<uint32_t> db(N * 32), de(N * 1024), dd(N * 1024);
    db.up(bases);
    Delta<uint32_t>::transpose_delta_pack_device(32, dv.p, db.p, de.p, N);
    Delta<uint32_t>::undelta_pack_untranspose_device(32, de.p, db.p, dd.p, N);
    EXPECT(dd.down() == v);
    // == the three-step composition of the reference (delta.rs:88-95)
    DevVec<uint32_t> dt(N * 1024), ddel(N * 1024), dpk(N * 1024);
    Transpose<uint32_t>::transpose_device(dv.p, dt.p, N);
    Delta<uint32_t>::delta_device(dt.p, db.p, ddel.p, N);
    BitPacking<uint32_t>::pack_device(32, ddel.p, dpk.p, N);
    EXPECT(dpk.down() == de.down());
    // mixed widths: blocks alternate W = 7 and W = 9 (values fit both)
    std::vector<uint8_t> widths(N);
    for (size_t b = 0; b < N; ++b) widths[b] = (b % 2) ? 9 : 7;
    MixedWidthPlan<uint32_t> plan(widths.data(), N);
    EXPECT(plan.n_blocks() == N && plan.packed_bytes() == (N / 2) * 128 * (7 + 9));
    DevVec<uint32_t> dm(plan.packed_bytes() / 4), dmu(N * 1024);
    plan.pack_device(dv.p, dm.p);
    plan.unpack_device(dm.p, dmu.p);
    EXPECT(dmu.down() == v);
    // the same column through the device-resident widths[] / offsets[] surface (nothing built on the host)
    DevVec<uint8_t> dw(N);
    dw.up(widths);
    DevVec<uint64_t> doff(N), dtot(1);
    DevVec<uint32_t> derr(1), dm2(plan.packed_bytes() / 4), dmu2(N * 1024);
    (void)hipMemset(derr.p, 0, 4);
    widths_to_offsets_device<uint32_t>(dw.p, N, doff.p, dtot.p, derr.p);
    pack_widths_device<uint32_t>(dw.p, doff.p, dv.p, dm2.p, plan.packed_bytes(), N, derr.p);
    unpack_widths_device<uint32_t>(dw.p, doff.p, dm2.p, plan.packed_bytes(), dmu2.p, N, derr.p);
    EXPECT(dtot.down()[0] == plan.packed_bytes());
    EXPECT(dm2.down() == dm.down());
    EXPECT(dmu2.down() == v);
    EXPECT(derr.down()[0] == 0u);
    EXPECT(hipDeviceSynchronize() == hipSuccess);
    // the same blocks as MANY SMALL ARRAYS in one launch (device arrays of pointers): 4 arrays of N/4 blocks, widths 7 / 9 / 7 / 9
    // are not uniform per array here, so take arrays of ONE block each with that block's width -- N arrays
    std::vector<const uint32_t*> hp(N);
    std::vector<uint32_t*> ho(N);
    std::vector<uint64_t> hoff = doff.down();
    std::vector<uint32_t> hnb(N, 1u);
    DevVec<uint32_t> dmu3(N * 1024);
    for (size_t b = 0; b < N; ++b) { hp[b] = dm2.p + hoff[b] / 4; ho[b] = dmu3.p + b * 1024; }
    DevVec<const uint32_t*> dpp(N);
    DevVec<uint32_t*> dop(N);
    DevVec<uint32_t> dnb(N);
    dpp.up(hp); dop.up(ho); dnb.up(hnb);
    unpack_batch_device<uint32_t>(dpp.p, dop.p, dw.p, dnb.p, N, 1, derr.p);
    EXPECT(dmu3.down() == v);
    EXPECT(derr.down()[0] == 0u);
    EXPECT(hipDeviceSynchronize() == hipSuccess);
}

// The caller loop of benches/bitpacking.rs:80-97 as ONE checked call on device slices (fastlanes_amd.hpp: DeviceSlice,
// unpack_column / pack_column / undelta_pack_column / unfor_pack_column / unpack_chunks): results equal the per-block host-tier
// trait calls, and every length the reference's loop asserts per block (bitpacking.rs:78-80, :111-113) is refused up front.
template <typename F> static bool throws_length(F&& f)
{
    try { f(); } catch (const std::length_error&) { return true; } catch (...) { return false; }
    return false;
}
static void test_column_api()
{
    using T = uint32_t;
    const size_t N = 37, W = 11, PL = 1024 * W / 32, LANES = 32;
    std::vector<T> v(N * 1024), bases(N * LANES), refs(N);
    for (size_t i = 0; i < v.size(); ++i) v[i] = (T)((i * 2654435761ull) >> 5) & ((1u << W) - 1u);
    for (size_t i = 0; i < bases.size(); ++i) bases[i] = (T)(i * 40503u + 7u);
    for (size_t b = 0; b < N; ++b) refs[b] = (T)(1000003u * (b + 1));
    DevVec<T> dv(N * 1024), dp(N * PL), du(N * 1024), db(N * LANES), dr(N);
    dv.up(v); db.up(bases); dr.up(refs);
    const DeviceSlice<T> sv(dv.p, dv.n), sp(dp.p, dp.n), su(du.p, du.n);
    const DeviceSlice<const T> cb(db.p, db.n), cr(dr.p, dr.n);
    pack_column<T>(W, sv, sp);
    unpack_column<T>(W, sp, su);
    EXPECT(du.down() == v);
    // == the loop of host-tier trait calls, block by block
    const std::vector<T> pk = dp.down();
    for (size_t b = 0; b < N; b += 9) {
        std::vector<T> one(PL);
        BitPacking<T>::unchecked_pack(W, v.data() + b * 1024, 1024, one.data(), PL);
        EXPECT(std::memcmp(one.data(), pk.data() + b * PL, PL * sizeof(T)) == 0);
    }
    // the same round trip inside buffers from the optional allocation helper (fl_column_pair_alloc), every layout
    for (int layout : {FL_LAYOUT_SEPARATE, FL_LAYOUT_ZONED, FL_LAYOUT_PROBE, FL_LAYOUT_INTERLEAVED}) {
        ColumnPair<T> pair(N * PL, N * 1024, 0, layout);
        EXPECT(pair.in().len == N * PL && pair.out().len == N * 1024 && pair.aux().ptr == nullptr);
        EXPECT(pair.layout() == FL_LAYOUT_SEPARATE || pair.layout() == FL_LAYOUT_ZONED || pair.layout() == FL_LAYOUT_INTERLEAVED);
        EXPECT(layout == FL_LAYOUT_PROBE || pair.layout() == layout);
        EXPECT(hipMemcpy(pair.in().ptr, dp.p, N * PL * sizeof(T), hipMemcpyDeviceToDevice) == hipSuccess);
        unpack_column<T>(W, DeviceSlice<const T>(pair.in()), pair.out());
        std::vector<T> back(N * 1024);
        EXPECT(hipMemcpy(back.data(), pair.out().ptr, back.size() * sizeof(T), hipMemcpyDeviceToHost) == hipSuccess);
        EXPECT(back == v);
    }
    // fused Delta / FoR decode of the column == the per-block trait calls
    undelta_pack_column<T>(W, sp, cb, su);
    std::vector<T> got = du.down();
    for (size_t b = 0; b < N; b += 12) {
        T in1[PL], base1[LANES], out1[1024];
        std::memcpy(in1, pk.data() + b * PL, sizeof in1);
        std::memcpy(base1, bases.data() + b * LANES, sizeof base1);
        Delta<T>::undelta_pack<W>(in1, base1, out1);
        EXPECT(std::memcmp(out1, got.data() + b * 1024, sizeof out1) == 0);
    }
    unfor_pack_column<T>(W, sp, cr, su);
    got = du.down();
    for (size_t b = 0; b < N; b += 12) {
        T in1[PL], out1[1024];
        std::memcpy(in1, pk.data() + b * PL, sizeof in1);
        FoR<T>::unfor_pack<W>(in1, refs[b], out1);
        EXPECT(std::memcmp(out1, got.data() + b * 1024, sizeof out1) == 0);
    }
    unfor_pack_column<T>(W, sp, cr.subslice(3, 1), su);                     // one reference for the whole column
    got = du.down();
    for (size_t i = 0; i < got.size(); i += 401) EXPECT(got[i] == (T)(v[i] + refs[3]));
    // a sharded caller hands each device / stream a sub-slice: blocks [10, 25)
    DevVec<T> du2(15 * 1024);
    unpack_column<T>(W, DeviceSlice<const T>(sp).subslice(10 * PL, 15 * PL), DeviceSlice<T>(du2.p, du2.n));
    EXPECT(du2.down() == std::vector<T>(v.begin() + 10 * 1024, v.begin() + 25 * 1024));
    // every length the per-block loop asserts (bitpacking.rs:78-80, :111-113) is refused BEFORE anything is launched
    EXPECT(throws_length([&] { unpack_column<T>(W, sp.subslice(0, sp.len - 1), su); }));           // packed one element short
    EXPECT(throws_length([&] { unpack_column<T>(W, sp, su.subslice(0, su.len - 1)); }));           // output not whole blocks
    EXPECT(throws_length([&] { unpack_column<T>(W, sp, su.subslice(0, su.len - 1024)); }));        // one block fewer than packed
    EXPECT(throws_length([&] { unpack_column<T>(W + 1, sp, su); }));                               // the width the data was not packed with
    EXPECT(throws_length([&] { pack_column<T>(W, sv, sp.subslice(0, sp.len - PL)); }));
    EXPECT(throws_length([&] { undelta_pack_column<T>(W, sp, cb.subslice(0, cb.len - 1), su); })); // bases: LANES per block
    EXPECT(throws_length([&] { unfor_pack_column<T>(W, sp, cr.subslice(0, 2), su); }));            // neither one reference nor one per block
    try { unpack_column<T>(33, sp, su); EXPECT(false); } catch (const Error& e) { EXPECT(e.status == FL_ERR_WIDTH); }   // bitpacking.rs:126
    try { (void)sp.subslice(sp.len, 1); EXPECT(false); } catch (const std::out_of_range&) {}
    unpack_column<T>(0, DeviceSlice<const T>(), DeviceSlice<T>());                                 // an empty column is a no-op
    // many small arrays ("chunks") in one launch through the checked table: chunks of 5, 0, 7 and 25 blocks of the column above
    const size_t cb_[4] = {5, 0, 7, 25}, first[4] = {0, 5, 5, 12};
    std::vector<const T*> hp(4);
    std::vector<T*> ho(4);
    std::vector<uint8_t> hw(4, (uint8_t)W);
    std::vector<uint32_t> hn(4);
    DevVec<T> du3(N * 1024);
    (void)hipMemset(du3.p, 0, du3.n * sizeof(T));
    for (int c = 0; c < 4; ++c) { hp[c] = dp.p + first[c] * PL; ho[c] = du3.p + first[c] * 1024; hn[c] = (uint32_t)cb_[c]; }
    DevVec<const T*> dpp(4);
    DevVec<T*> dop(4);
    DevVec<uint8_t> dw(4);
    DevVec<uint32_t> dn(4), derr(1);
    dpp.up(hp); dop.up(ho); dw.up(hw); dn.up(hn);
    (void)hipMemset(derr.p, 0, 4);
    ChunkTable<T> t;
    t.packed = DeviceSlice<const T* const>(dpp.p, 4);
    t.out = DeviceSlice<T* const>(dop.p, 4);
    t.widths = DeviceSlice<const uint8_t>(dw.p, 4);
    t.n_blocks = DeviceSlice<const uint32_t>(dn.p, 4);
    t.max_blocks = 25;
    unpack_chunks<T>(t, derr.p);
    EXPECT(du3.down() == v);
    EXPECT(derr.down()[0] == 0u);
    t.n_blocks = DeviceSlice<const uint32_t>(dn.p, 3);                       // a table whose arrays disagree is refused on the host
    EXPECT(throws_length([&] { unpack_chunks<T>(t, derr.p); }));
    EXPECT(hipDeviceSynchronize() == hipSuccess);
}

int main()
{
    try {
        test_ordering_is_own_inverse();
        pack_u16_into_u3_no_unsafe();
        test_unchecked_pack();
        test_unpack_single();
        test_round_trip_matrix();
        test_delta();
        test_ffor();
        test_panics();
        test_wide_and_narrow();
        test_device_tier();
        test_column_api();
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 2;
    }
    std::printf(failures ? "FAILED (%d)\n" : "ok\n", failures);
    return failures ? 1 : 0;
}
