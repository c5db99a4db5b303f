// fl_capi.hip -- the extern "C" boundary declared in include/fastlanes_amd.h.
// Validates arguments, maps the runtime width to the per-(T,W) kernel instance
// (the reference's `match width`, bitpacking.rs:82-95) and launches it.  No CPU
// compute path exists in this library: every entry point ends in a HIP launch.
#include "../../include/fastlanes_amd.h"
#include "fl_kernels.hpp"
#include "fl_misc.hpp"
#include "fl_mixed.hpp"
#include "fl_consume.hpp"

#include <algorithm>
#include <new>
#include <vector>

namespace {

using namespace fl;

thread_local int g_last_hip_error = 0;

inline int hip_fail(hipError_t e)
{
    g_last_hip_error = (int)e;
    return FL_ERR_HIP;
}

inline bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }

template <typename T>
int run_stream(stream_launch_t fn, const T* in, T* out, const void* aux, size_t aux_stride,
               size_t n_blocks, bool need_in, bool need_out, bool need_aux, void* stream)
{
    if (n_blocks == 0) return FL_OK;
    if ((need_in && !in) || (need_out && !out) || (need_aux && !aux)) return FL_ERR_NULL;
    if (misaligned(in) || misaligned(out)) return FL_ERR_ALIGN;
    StreamArgs a;
    a.in = reinterpret_cast<const u32x4*>(in);
    a.out = reinterpret_cast<u32x4*>(out);
    a.aux = aux;
    a.aux_stride = aux_stride;
    a.n_blocks = n_blocks;
    a.tiles_per_xcd = 0;   // filled by the launcher
    hipError_t e = fn(a, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? FL_OK : hip_fail(e);
}

template <typename T> int dev_pack(unsigned w, const T* in, T* out, size_t n, void* s)
{
    if (w > (unsigned)Elem<T>::BITS) return FL_ERR_WIDTH;
    return run_stream<T>(pack_table_impl<T, PACK_PLAIN>().fn[w], in, out, nullptr, 0, n, true, w != 0, false, s);
}
template <typename T> int dev_unpack(unsigned w, const T* in, T* out, size_t n, void* s)
{
    if (w > (unsigned)Elem<T>::BITS) return FL_ERR_WIDTH;
    return run_stream<T>(unpack_table_impl<T, BODY_STORE>().fn[w], in, out, nullptr, 0, n, w != 0, true, false, s);
}
template <typename T>
int dev_for_pack(unsigned w, const T* in, const T* refs, size_t stride, T* out, size_t n, void* s)
{
    if (w > (unsigned)Elem<T>::BITS) return FL_ERR_WIDTH;
    return run_stream<T>(pack_table_impl<T, PACK_FOR>().fn[w], in, out, refs, stride, n, true, w != 0, true, s);
}
template <typename T>
int dev_unfor_pack(unsigned w, const T* in, const T* refs, size_t stride, T* out, size_t n, void* s)
{
    if (w > (unsigned)Elem<T>::BITS) return FL_ERR_WIDTH;
    return run_stream<T>(unpack_table_impl<T, BODY_ADD_REF>().fn[w], in, out, refs, stride, n, w != 0, true, true, s);
}
template <typename T>
int dev_undelta_pack(unsigned w, const T* in, const T* bases, T* out, size_t n, void* s)
{
    if (w > (unsigned)Elem<T>::BITS) return FL_ERR_WIDTH;
    if (n && misaligned(bases)) return FL_ERR_ALIGN;
    return run_stream<T>(unpack_table_impl<T, BODY_UNDELTA>().fn[w], in, out, bases, 0, n, w != 0, true, true, s);
}
template <typename T>
int dev_undelta_pack_untranspose(unsigned w, const T* in, const T* bases, T* out, size_t n, void* s)
{
    if (w > (unsigned)Elem<T>::BITS) return FL_ERR_WIDTH;
    if (n && misaligned(bases)) return FL_ERR_ALIGN;
    return run_stream<T>(unpack_table_impl<T, BODY_UNDELTA_UNTRANSPOSE>().fn[w], in, out, bases, 0, n, w != 0, true, true, s);
}
template <typename T>
int dev_transpose_delta_pack(unsigned w, const T* in, const T* bases, T* out, size_t n, void* s)
{
    if (w > (unsigned)Elem<T>::BITS) return FL_ERR_WIDTH;
    if (n && misaligned(bases)) return FL_ERR_ALIGN;
    return run_stream<T>(pack_table_impl<T, PACK_TRANSPOSE_DELTA>().fn[w], in, out, bases, 0, n, true, w != 0, true, s);
}
template <typename T>
int dev_unpack_block_sums(unsigned w, const T* in, size_t n, uint64_t* sums, void* s)
{
    if (w > (unsigned)Elem<T>::BITS) return FL_ERR_WIDTH;
    if (n == 0) return FL_OK;
    if (!sums || (w != 0 && !in)) return FL_ERR_NULL;
    if (misaligned(in)) return FL_ERR_ALIGN;
    ReduceArgs a{reinterpret_cast<const u32x4*>(in), sums, nullptr, n, 0};
    hipError_t e = sum_table_impl<T>().fn[w](a, static_cast<hipStream_t>(s));
    return e == hipSuccess ? FL_OK : hip_fail(e);
}
template <typename T>
int dev_unpack_compare(unsigned w, const T* in, int op, T constant, size_t n, uint32_t* mask, void* s)
{
    if (w > (unsigned)Elem<T>::BITS) return FL_ERR_WIDTH;
    if (op < FL_CMP_EQ || op > FL_CMP_GE) return FL_ERR_INDEX;
    if (n == 0) return FL_OK;
    if (!mask || (w != 0 && !in)) return FL_ERR_NULL;
    if (misaligned(in) || misaligned(mask)) return FL_ERR_ALIGN;
    // reduce the six predicates to  x == k  /  x <= k  plus a complement
    const T MAXV = (T) ~(T)0;
    CompareArgs a;
    a.in = reinterpret_cast<const u32x4*>(in);
    a.mask = reinterpret_cast<u32x4*>(mask);
    a.n_blocks = n;
    a.tiles_per_xcd = 0;
    a.is_eq = 0;
    a.invert = 0;
    a.constant = constant;
    switch (op) {
    case FL_CMP_EQ: a.is_eq = 1; break;
    case FL_CMP_NE: a.is_eq = 1; a.invert = 1; break;
    case FL_CMP_LE: break;
    case FL_CMP_GT: a.invert = 1; break;
    case FL_CMP_LT:                       // x < k  ==  x <= k-1 ;  x < 0 is never true
        if (constant == 0) { a.constant = MAXV; a.invert = 1; } else a.constant = (T)(constant - 1);
        break;
    default:                              // FL_CMP_GE: x >= k == !(x <= k-1) ; x >= 0 is always true
        if (constant == 0) a.constant = MAXV; else { a.constant = (T)(constant - 1); a.invert = 1; }
        break;
    }
    hipError_t e = (a.is_eq ? compare_table_impl<T, true>() : compare_table_impl<T, false>()).fn[w](a, static_cast<hipStream_t>(s));
    return e == hipSuccess ? FL_OK : hip_fail(e);
}
template <typename T>
int dev_block_min_max(const T* in, size_t n, T* mins, T* maxs, void* s)
{
    if (n == 0) return FL_OK;
    if (!in || !mins || !maxs) return FL_ERR_NULL;
    if (misaligned(in)) return FL_ERR_ALIGN;
    ReduceArgs a{reinterpret_cast<const u32x4*>(in), mins, maxs, n, 0};
    hipError_t e = min_max_launcher<T>()(a, static_cast<hipStream_t>(s));
    return e == hipSuccess ? FL_OK : hip_fail(e);
}
template <typename T> int dev_delta(bool inverse, const T* in, const T* bases, T* out, size_t n, void* s)
{
    if (n && misaligned(bases)) return FL_ERR_ALIGN;
    return run_stream<T>(delta_launcher<T>(inverse), in, out, bases, 0, n, true, true, true, s);
}
template <typename T> int dev_transpose(bool inverse, const T* in, T* out, size_t n, void* s)
{
    return run_stream<T>(transpose_launcher<T>(inverse), in, out, nullptr, 0, n, true, true, false, s);
}
template <typename T>
int dev_unpack_single(unsigned w, const T* packed, size_t n_blocks, const uint64_t* idx, size_t n_idx,
                      T* out, uint32_t* err_flag, void* s)
{
    if (w > (unsigned)Elem<T>::BITS) return FL_ERR_WIDTH;
    if (n_idx == 0) return FL_OK;
    if (!idx || !out || (w != 0 && !packed)) return FL_ERR_NULL;
    SingleArgs a{packed, idx, out, err_flag, n_blocks, n_idx, w};
    hipError_t e = unpack_single_launch<T>(a, static_cast<hipStream_t>(s));
    return e == hipSuccess ? FL_OK : hip_fail(e);
}

// ---------------------------------------------------------------------------
// Host tier: stage host slices through device memory and run the same kernels.
// ---------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    hipError_t alloc(size_t bytes) { return bytes ? hipMalloc(&p, bytes) : hipSuccess; }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

#define FL_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return hip_fail(e_); } while (0)

template <typename T, typename F>
int host_run(const T* in, size_t in_elems, const T* aux, size_t aux_elems, T* out, size_t out_elems, F&& dev)
{
    if ((in_elems && !in) || (out_elems && !out) || (aux_elems && !aux)) return FL_ERR_NULL;
    DevBuf din, daux, dout;
    FL_HIP(din.alloc(in_elems * sizeof(T)));
    FL_HIP(daux.alloc(aux_elems * sizeof(T)));
    FL_HIP(dout.alloc(out_elems * sizeof(T)));
    if (in_elems) FL_HIP(hipMemcpy(din.p, in, in_elems * sizeof(T), hipMemcpyHostToDevice));
    if (aux_elems) FL_HIP(hipMemcpy(daux.p, aux, aux_elems * sizeof(T), hipMemcpyHostToDevice));
    int rc = dev(static_cast<const T*>(din.p), static_cast<const T*>(daux.p), static_cast<T*>(dout.p));
    if (rc != FL_OK) return rc;
    FL_HIP(hipStreamSynchronize(nullptr));
    if (out_elems) FL_HIP(hipMemcpy(out, dout.p, out_elems * sizeof(T), hipMemcpyDeviceToHost));
    return FL_OK;
}

template <typename T> size_t plen(unsigned w) { return (size_t)1024 * w / Elem<T>::BITS; }

}  // namespace

// ---------------------------------------------------------------------------
// mixed-width plan
// ---------------------------------------------------------------------------
struct fl_mixed_plan {
    unsigned type_bits = 0;
    size_t n_blocks = 0;
    uint64_t packed_bytes = 0;
    uint64_t n_tiles = 0;
    MixedEntry* d_entries = nullptr; // (block, packed offset) bucketed by width, ascending inside a bucket
    uint64_t* d_offsets = nullptr;   // byte offset of every block in the packed column, natural order
    MixedTile* d_tiles = nullptr;    // one descriptor per tile of <= 32 same-width blocks
    bool window_unpack = false;      // every tile's span fits a 32-bit store window
    bool window_pack = false;
};

namespace {

template <typename T>
int run_mixed(bool pack, const fl_mixed_plan* p, const void* packed, void* unpacked, void* stream)
{
    if (!p) return FL_ERR_NULL;
    if (p->type_bits != (unsigned)Elem<T>::BITS) return FL_ERR_WIDTH;
    if (p->n_blocks == 0) return FL_OK;
    if (!unpacked || (p->packed_bytes && !packed)) return FL_ERR_NULL;
    if (misaligned(packed) || misaligned(unpacked)) return FL_ERR_ALIGN;
    MixedArgs a;
    a.packed = static_cast<const char*>(packed);
    a.unpacked = static_cast<char*>(unpacked);
    a.entries = p->d_entries;
    a.tiles = p->d_tiles;
    a.n_tiles = p->n_tiles;
    a.tiles_per_xcd = 0;
    const bool window = pack ? p->window_pack : p->window_unpack;
    hipError_t e = (pack ? mixed_pack_launcher<T>(window) : mixed_unpack_launcher<T>(window))(a, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? FL_OK : hip_fail(e);
}

}  // namespace

extern "C" {

static int mixed_plan_create_impl(unsigned type_bits, const uint8_t* widths, size_t n_blocks, fl_mixed_plan** plan)
{
    if (!plan || (n_blocks && !widths)) return FL_ERR_NULL;
    *plan = nullptr;
    if (type_bits != 8 && type_bits != 16 && type_bits != 32 && type_bits != 64) return FL_ERR_WIDTH;
    if (n_blocks > 0xFFFFFFFFull) return FL_ERR_INDEX;
    size_t count[66] = {0};
    for (size_t b = 0; b < n_blocks; ++b) {
        if (widths[b] > type_bits) return FL_ERR_WIDTH;   // bitpacking.rs:93 unreachable!()
        ++count[widths[b] + 1];
    }
    fl_mixed_plan* p = new (std::nothrow) fl_mixed_plan;
    if (!p) return FL_ERR_HIP;
    struct Guard {   // frees the half-built plan on every early exit, including exceptions
        fl_mixed_plan* p;
        ~Guard() { if (p) fl_mixed_plan_destroy(p); }
    } guard{p};
    p->type_bits = type_bits;
    p->n_blocks = n_blocks;
    size_t bucket_start[67] = {0};
    for (unsigned w = 0; w <= 64; ++w) bucket_start[w + 1] = bucket_start[w] + count[w + 1];
    std::vector<uint32_t> ids(n_blocks);
    std::vector<uint64_t> off(n_blocks);
    {
        size_t cursor[66];
        for (unsigned w = 0; w <= 64; ++w) cursor[w] = bucket_start[w];
        uint64_t o = 0;
        for (size_t b = 0; b < n_blocks; ++b) {
            off[b] = o;
            o += 128ull * widths[b];
            ids[cursor[widths[b]]++] = (uint32_t)b;
        }
        p->packed_bytes = o;
    }
    // tiles: 32 same-width blocks each, buckets in width order; check the store windows
    const uint64_t block_bytes = 128ull * type_bits;
    std::vector<MixedEntry> entries(n_blocks);
    for (size_t i = 0; i < n_blocks; ++i) entries[i] = MixedEntry{ids[i], off[ids[i]]};
    std::vector<MixedTile> tiles;
    tiles.reserve(n_blocks / 32 + 66);
    bool wu = true, wp = true;
    for (unsigned w = 0; w <= type_bits; ++w) {
        const size_t s0 = bucket_start[w], m = bucket_start[w + 1] - s0;
        for (size_t t = 0; t * 32 < m; ++t) {
            const unsigned cnt = (unsigned)(m - t * 32 < 32 ? m - t * 32 : 32);
            const uint64_t first = ids[s0 + t * 32], last = ids[s0 + t * 32 + cnt - 1];
            tiles.push_back(MixedTile{(uint32_t)(s0 + t * 32), cnt | (w << 8), first, off[first], 0});
            if ((last - first + 1) * block_bytes > 0xFFFFFFFFull) wu = false;
            if (off[last] + 128ull * w - off[first] > 0xFFFFFFFFull) wp = false;
        }
    }
    // Launch order = column order: tiles are sorted by their first block, so neighbouring
    // workgroups (and each XCD's contiguous share of the tile list) cover the same region of
    // the column whatever their widths -- the global traffic stays a dense sweep.  (Bucket
    // order, i.e. one width after the other, measured 20-25 % slower on interleaved widths.)
    std::sort(tiles.begin(), tiles.end(), [](const MixedTile& x, const MixedTile& y) { return x.first_blk < y.first_blk; });
    p->n_tiles = tiles.size();
    p->window_unpack = wu;
    p->window_pack = wp;
    if (n_blocks) {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p->d_entries), n_blocks * sizeof(MixedEntry));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_offsets), n_blocks * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_tiles), tiles.size() * sizeof(MixedTile));
        if (e == hipSuccess) e = hipMemcpy(p->d_entries, entries.data(), n_blocks * sizeof(MixedEntry), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(p->d_offsets, off.data(), n_blocks * sizeof(uint64_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(p->d_tiles, tiles.data(), tiles.size() * sizeof(MixedTile), hipMemcpyHostToDevice);
        if (e != hipSuccess) return hip_fail(e);
    }
    guard.p = nullptr;
    *plan = p;
    return FL_OK;
}

int fl_mixed_plan_create(unsigned type_bits, const uint8_t* widths, size_t n_blocks, fl_mixed_plan** plan)
{
    // nothing may throw across the C ABI (the host-side tables are std::vectors)
    try {
        return mixed_plan_create_impl(type_bits, widths, n_blocks, plan);
    } catch (...) {
        g_last_hip_error = (int)hipErrorOutOfMemory;
        return FL_ERR_HIP;
    }
}

void fl_mixed_plan_destroy(fl_mixed_plan* p)
{
    if (!p) return;
    if (p->d_entries) (void)hipFree(p->d_entries);
    if (p->d_offsets) (void)hipFree(p->d_offsets);
    if (p->d_tiles) (void)hipFree(p->d_tiles);
    delete p;
}
size_t fl_mixed_plan_n_blocks(const fl_mixed_plan* p) { return p ? p->n_blocks : 0; }
uint64_t fl_mixed_plan_packed_bytes(const fl_mixed_plan* p) { return p ? p->packed_bytes : 0; }
const uint64_t* fl_mixed_plan_offsets(const fl_mixed_plan* p) { return p ? p->d_offsets : nullptr; }

const char* fl_version(void) { return "fastlanes_amd 0.1.0 (gfx950; wire format of spiraldb/fastlanes 0.1.8)"; }

const char* fl_status_string(int status)
{
    switch (status) {
    case FL_OK: return "ok";
    case FL_ERR_WIDTH: return "width > T";
    case FL_ERR_INDEX: return "index out of range";
    case FL_ERR_NULL: return "null pointer";
    case FL_ERR_ALIGN: return "device pointer not 16-byte aligned";
    case FL_ERR_HIP: return "HIP runtime error";
    default: return "unknown status";
    }
}

int fl_last_hip_error(void) { return g_last_hip_error; }

size_t fl_packed_len(unsigned type_bits, unsigned width)
{
    if (type_bits != 8 && type_bits != 16 && type_bits != 32 && type_bits != 64) return 0;
    if (width > type_bits) return 0;
    return (size_t)1024 * width / type_bits;
}

#define FL_DEFINE_TYPE(T, S)                                                                              \
    int fl_##S##_pack(unsigned w, const T* in, T* out, size_t n, void* s) { return dev_pack<T>(w, in, out, n, s); } \
    int fl_##S##_unpack(unsigned w, const T* in, T* out, size_t n, void* s) { return dev_unpack<T>(w, in, out, n, s); } \
    int fl_##S##_unpack_single(unsigned w, const T* pk, size_t n, const uint64_t* idx, size_t ni, T* out,  \
                               uint32_t* ef, void* s)                                                     \
    { return dev_unpack_single<T>(w, pk, n, idx, ni, out, ef, s); }                                       \
    int fl_##S##_for_pack(unsigned w, const T* in, const T* r, size_t rs, T* out, size_t n, void* s)      \
    { return dev_for_pack<T>(w, in, r, rs, out, n, s); }                                                  \
    int fl_##S##_unfor_pack(unsigned w, const T* in, const T* r, size_t rs, T* out, size_t n, void* s)    \
    { return dev_unfor_pack<T>(w, in, r, rs, out, n, s); }                                                \
    int fl_##S##_delta(const T* in, const T* b, T* out, size_t n, void* s) { return dev_delta<T>(false, in, b, out, n, s); } \
    int fl_##S##_undelta(const T* in, const T* b, T* out, size_t n, void* s) { return dev_delta<T>(true, in, b, out, n, s); } \
    int fl_##S##_undelta_pack(unsigned w, const T* in, const T* b, T* out, size_t n, void* s)             \
    { return dev_undelta_pack<T>(w, in, b, out, n, s); }                                                  \
    int fl_##S##_undelta_pack_untranspose(unsigned w, const T* in, const T* b, T* out, size_t n, void* s) \
    { return dev_undelta_pack_untranspose<T>(w, in, b, out, n, s); }                                      \
    int fl_##S##_transpose_delta_pack(unsigned w, const T* in, const T* b, T* out, size_t n, void* s)     \
    { return dev_transpose_delta_pack<T>(w, in, b, out, n, s); }                                          \
    int fl_##S##_unpack_block_sums(unsigned w, const T* in, size_t n, uint64_t* sums, void* s)           \
    { return dev_unpack_block_sums<T>(w, in, n, sums, s); }                                               \
    int fl_##S##_unpack_compare(unsigned w, const T* in, int op, T k, size_t n, uint32_t* mask, void* s)  \
    { return dev_unpack_compare<T>(w, in, op, k, n, mask, s); }                                           \
    int fl_##S##_block_min_max(const T* in, size_t n, T* mins, T* maxs, void* s)                          \
    { return dev_block_min_max<T>(in, n, mins, maxs, s); }                                                \
    int fl_##S##_transpose(const T* in, T* out, size_t n, void* s) { return dev_transpose<T>(false, in, out, n, s); } \
    int fl_##S##_untranspose(const T* in, T* out, size_t n, void* s) { return dev_transpose<T>(true, in, out, n, s); } \
    int fl_##S##_unpack_mixed(const fl_mixed_plan* p, const T* pk, T* out, void* s) { return run_mixed<T>(false, p, pk, out, s); } \
    int fl_##S##_pack_mixed(const fl_mixed_plan* p, const T* in, T* pk, void* s) { return run_mixed<T>(true, p, pk, const_cast<T*>(in), s); } \
    int fl_##S##_pack_host(unsigned w, const T* in, T* out, size_t n)                                     \
    {                                                                                                     \
        if (w > sizeof(T) * 8) return FL_ERR_WIDTH;                                                       \
        return host_run<T>(in, n * 1024, nullptr, 0, out, n * plen<T>(w),                                 \
                           [&](const T* di, const T*, T* d_o) { return dev_pack<T>(w, di, d_o, n, nullptr); }); \
    }                                                                                                     \
    int fl_##S##_unpack_host(unsigned w, const T* in, T* out, size_t n)                                   \
    {                                                                                                     \
        if (w > sizeof(T) * 8) return FL_ERR_WIDTH;                                                       \
        return host_run<T>(in, n * plen<T>(w), nullptr, 0, out, n * 1024,                                 \
                           [&](const T* di, const T*, T* d_o) { return dev_unpack<T>(w, di, d_o, n, nullptr); }); \
    }                                                                                                     \
    int fl_##S##_unpack_single_host(unsigned w, const T* pk, size_t n, uint64_t index, T* value)          \
    {                                                                                                     \
        if (w > sizeof(T) * 8) return FL_ERR_WIDTH;                                                       \
        if (!value) return FL_ERR_NULL;                                                                   \
        if (w == 0) { *value = 0; return FL_OK; } /* bitpacking.rs:136-139 precedes the assert */         \
        if (index >= (uint64_t)n * 1024) return FL_ERR_INDEX;                                             \
        DevBuf didx;                                                                                      \
        FL_HIP(didx.alloc(sizeof(uint64_t)));                                                             \
        FL_HIP(hipMemcpy(didx.p, &index, sizeof(uint64_t), hipMemcpyHostToDevice));                       \
        return host_run<T>(pk, n * plen<T>(w), nullptr, 0, value, 1, [&](const T* di, const T*, T* d_o) {  \
            return dev_unpack_single<T>(w, di, n, static_cast<const uint64_t*>(didx.p), 1, d_o, nullptr, nullptr); \
        });                                                                                               \
    }                                                                                                     \
    int fl_##S##_for_pack_host(unsigned w, const T* in, T reference, T* out, size_t n)                    \
    {                                                                                                     \
        if (w > sizeof(T) * 8) return FL_ERR_WIDTH;                                                       \
        return host_run<T>(in, n * 1024, &reference, 1, out, n * plen<T>(w),                              \
                           [&](const T* di, const T* da, T* d_o) { return dev_for_pack<T>(w, di, da, 0, d_o, n, nullptr); }); \
    }                                                                                                     \
    int fl_##S##_unfor_pack_host(unsigned w, const T* in, T reference, T* out, size_t n)                  \
    {                                                                                                     \
        if (w > sizeof(T) * 8) return FL_ERR_WIDTH;                                                       \
        return host_run<T>(in, n * plen<T>(w), &reference, 1, out, n * 1024,                              \
                           [&](const T* di, const T* da, T* d_o) { return dev_unfor_pack<T>(w, di, da, 0, d_o, n, nullptr); }); \
    }                                                                                                     \
    int fl_##S##_delta_host(const T* in, const T* b, T* out, size_t n)                                    \
    {                                                                                                     \
        return host_run<T>(in, n * 1024, b, n * (1024 / (sizeof(T) * 8)), out, n * 1024,                  \
                           [&](const T* di, const T* da, T* d_o) { return dev_delta<T>(false, di, da, d_o, n, nullptr); }); \
    }                                                                                                     \
    int fl_##S##_undelta_host(const T* in, const T* b, T* out, size_t n)                                  \
    {                                                                                                     \
        return host_run<T>(in, n * 1024, b, n * (1024 / (sizeof(T) * 8)), out, n * 1024,                  \
                           [&](const T* di, const T* da, T* d_o) { return dev_delta<T>(true, di, da, d_o, n, nullptr); }); \
    }                                                                                                     \
    int fl_##S##_undelta_pack_host(unsigned w, const T* in, const T* b, T* out, size_t n)                 \
    {                                                                                                     \
        if (w > sizeof(T) * 8) return FL_ERR_WIDTH;                                                       \
        return host_run<T>(in, n * plen<T>(w), b, n * (1024 / (sizeof(T) * 8)), out, n * 1024,            \
                           [&](const T* di, const T* da, T* d_o) { return dev_undelta_pack<T>(w, di, da, d_o, n, nullptr); }); \
    }                                                                                                     \
    int fl_##S##_transpose_host(const T* in, T* out, size_t n)                                            \
    {                                                                                                     \
        return host_run<T>(in, n * 1024, nullptr, 0, out, n * 1024,                                       \
                           [&](const T* di, const T*, T* d_o) { return dev_transpose<T>(false, di, d_o, n, nullptr); }); \
    }                                                                                                     \
    int fl_##S##_untranspose_host(const T* in, T* out, size_t n)                                          \
    {                                                                                                     \
        return host_run<T>(in, n * 1024, nullptr, 0, out, n * 1024,                                       \
                           [&](const T* di, const T*, T* d_o) { return dev_transpose<T>(true, di, d_o, n, nullptr); }); \
    }

FL_DEFINE_TYPE(uint8_t, u8)
FL_DEFINE_TYPE(uint16_t, u16)
FL_DEFINE_TYPE(uint32_t, u32)
FL_DEFINE_TYPE(uint64_t, u64)

}  // extern "C"
