// fl_capi.hip -- the extern "C" boundary declared in include/fastlanes_amd.h.
// Validates arguments, maps the runtime width to the per-(T,W) kernel instance
// (the reference's `match width`, bitpacking.rs:82-95) and launches it.  No CPU
// compute path exists in this library: every entry point ends in a HIP launch.
#include "../../include/fastlanes_amd.h"
#include "../../include/fastlanes_amd_internal.h"
#include "fl_kernels.hpp"
#include "fl_misc.hpp"
#include "fl_widths.hpp"
#include "fl_chain.hpp"
#include "fl_stream.hpp"
#include "fl_batch.hpp"
#include "fl_scan.hpp"
#include "fl_consume.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <mutex>
#include <new>
#include <cmath>
#include <vector>

namespace {

using namespace fl;

thread_local int g_last_hip_error = 0;

// fl_internal_set_kernel_policy: 0 = the generated table (fl_dispatch.hpp), 1 = cell-column kernels wherever they are
// built, 2 = wave-per-block kernels wherever they exist.  Results are bit-identical; only speed differs.
std::atomic<int> g_kernel_policy{0};

// waves per SIMD to run the wave-per-block kernel at, or 0 = use the cell-column kernel.  The per-(T,W) cell-column
// families are only built where the table chose them (cell_column_built); Delta's / Transpose's per-type cell-column
// kernels (no width parameter) always exist.
inline int chosen_waves(unsigned type_bits, unsigned w, fl::WaveOp op)
{
    const int p = g_kernel_policy.load(std::memory_order_relaxed);
    const int table = fl::wave_policy(type_bits, w, op);
    const bool per_type = op == fl::WAVE_UNDELTA || op == fl::WAVE_DELTA || op == fl::WAVE_TRANSPOSE || op == fl::WAVE_UNTRANSPOSE;
    if ((p & 0xff) == 1) return (per_type || fl::cell_column_built(type_bits, w, op)) ? 0 : table;
    if ((p & 0xff) != 2) return table;
    if ((p >> 8) & 0xff) return (p >> 8) & 0xff;     // A/B tools: policy 2 + 256 * waves forces the occupancy too
    return table ? table : fl::wave_fallback(type_bits, op == fl::WAVE_PACK || op == fl::WAVE_FOR_PACK || op == fl::WAVE_TRANSPOSE_DELTA_PACK);
}



inline int hip_fail(hipError_t e)
{
    g_last_hip_error = (int)e;
    return FL_ERR_HIP;
}

bool fl_in_constructed_pair(std::initializer_list<const void*> ptrs);   // below, next to fl_column_pair_alloc
inline bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }

// FL_CHECK_DEVICE=1 (fastlanes_amd.h "Threading and device selection"): before a device-tier launch, every pointer must be
// memory the calling thread's CURRENT device can use -- its own HBM, managed memory, or pinned host memory -- and `stream` a
// stream of that device; FL_ERR_DEVICE otherwise.  Off by default (one relaxed load per call): a raw kernel launch does not
// check either, and hipPointerGetAttributes costs microseconds.
inline bool device_check_enabled()
{
    static const bool on = [] { const char* e = getenv("FL_CHECK_DEVICE"); return e && e[0] && strcmp(e, "0") != 0; }();
    return on;
}
int device_check(void* stream, std::initializer_list<const void*> ptrs)
{
    if (!device_check_enabled()) return FL_OK;
    int cur = -1;
    if (hipError_t e = hipGetDevice(&cur); e != hipSuccess) return hip_fail(e);
    if (stream) {
        int sdev = -1;
        if (hipStreamGetDevice(static_cast<hipStream_t>(stream), &sdev) != hipSuccess) { (void)hipGetLastError(); return FL_ERR_DEVICE; }
        if (sdev != cur) return FL_ERR_DEVICE;
    }
    for (const void* p : ptrs) {
        if (!p) continue;                                   // NULL is judged (FL_ERR_NULL or allowed) by the entry point itself
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return FL_ERR_DEVICE; }   // plain host memory
        const bool device_mem = at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged || at.type == hipMemoryTypeArray;
        if (device_mem && at.type != hipMemoryTypeManaged && at.device != cur) return FL_ERR_DEVICE;
        if (!device_mem && at.type != hipMemoryTypeHost) return FL_ERR_DEVICE;   // unregistered host memory
    }
    return FL_OK;
}
// (every device-tier entry also tells the launchers whether its buffers lie inside one live FL_LAYOUT_INTERLEAVED pair: fl_kernels.hpp,
// constructed_pair_this_thread -- one relaxed load while no such pair exists)
#define FL_DEVICE_TIER(stream, ...) do { if (const int rc_ = device_check(stream, {__VA_ARGS__})) return rc_; \
                                         fl::constructed_pair_this_thread() = fl_in_constructed_pair({__VA_ARGS__}); } while (0)

template <typename T>
int run_stream(stream_launch_t fn, const T* in, T* out, const void* aux, size_t aux_stride,
               size_t n_blocks, bool need_in, bool need_out, bool need_aux, void* stream)
{
    if (n_blocks == 0) return FL_OK;
    if ((need_in && !in) || (need_out && !out) || (need_aux && !aux)) return FL_ERR_NULL;
    if (misaligned(in) || misaligned(out)) return FL_ERR_ALIGN;
    // a cell-column instance exists only where the dispatch table sends calls to it (fl_kernels.hpp: unpack_entry / pack_entry);
    // chosen_waves() never selects a missing one, but a table / build mismatch must be an error, not a call through nullptr
    if (!fn) return hip_fail(hipErrorInvalidDeviceFunction);
    StreamArgs a;
    a.in = reinterpret_cast<const u32x4*>(in);
    a.out = reinterpret_cast<u32x4*>(out);
    a.aux = aux;
    a.aux_stride = aux_stride;
    a.n_blocks = n_blocks;
    a.tiles_per_xcd = 0;   // filled by the launcher
    a.window_shift = 63;   // filled by the launcher
    hipError_t e = fn(a, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? FL_OK : hip_fail(e);
}

// Delta's bodies and the transposes on the wave-per-block pipeline kernel (fl_chain.hpp).  Returns -1 when no such form
// exists for the op: the caller then uses the cell-column kernel.
// Mixed-width form (widths != nullptr; the three ops with a packed side only): per-block widths[] / offsets[] read and checked
// by the kernel, `w` unused.
template <typename T>
int run_chain(int op, int waves, unsigned w, const T* in, const T* bases, T* out, size_t n_blocks, void* stream,
              const uint8_t* widths = nullptr, const uint64_t* offsets = nullptr, size_t packed_bytes = 0, uint32_t* err_flag = nullptr,
              bool two_blocks = false)
{
    fl::chain_launch_t fn = widths ? fl::chain_widths_launcher<T>(op) : fl::chain_launcher<T>(op);
    if (two_blocks && !widths) {                            // the two-blocks-per-wavefront form, where it exists (fl_chain.hpp)
        if (const fl::chain_launch_t fn2 = fl::chain_launcher_two_blocks<T>(op)) fn = fn2;
    }
    if (!fn) return -1;
    if (n_blocks == 0) return FL_OK;
    const bool packed_in = op == fl::OP_UNDELTA_PACK || op == fl::OP_UNDELTA_PACK_UNTRANSPOSE;
    const bool packed_out = op == fl::OP_TRANSPOSE_DELTA_PACK;
    const bool needs_bases = op != fl::OP_TRANSPOSE && op != fl::OP_UNTRANSPOSE;
    if (widths) {
        // a column whose blocks all have width 0 has no packed bytes: its packed pointer may be NULL (run_widths)
        static const T no_bytes[16 / sizeof(T)] __attribute__((aligned(16))) = {0};
        if (packed_bytes == 0 && packed_in && !in) in = no_bytes;
        if (packed_bytes == 0 && packed_out && !out) out = const_cast<T*>(no_bytes);   // never written: every block is skipped or has W = 0
        if (!(packed_in || packed_out) || !offsets) return FL_ERR_NULL;
    }
    if ((!out && !(packed_out && w == 0 && !widths)) || (needs_bases && !bases) || (!(packed_in && w == 0 && !widths) && !in)) return FL_ERR_NULL;
    if (misaligned(in) || misaligned(out) || misaligned(bases)) return FL_ERR_ALIGN;
    fl::ChainArgs a;
    a.in = reinterpret_cast<const char*>(in);
    a.out = reinterpret_cast<char*>(out);
    a.bases = reinterpret_cast<const char*>(bases);
    a.n_blocks = n_blocks;
    a.tiles_per_xcd = 0;
    a.window_shift = 63;
    a.width = w;
    a.widths = widths;
    a.offsets = offsets;
    a.err_flag = err_flag;
    a.packed_bytes = packed_bytes;
    a.nt_from = fl::nt_read_from(Elem<T>::BITS);
    hipError_t e = fn(a, waves, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? FL_OK : hip_fail(e);
}

// Uniform-width call served by the wave-per-block kernels (fl_dispatch.hpp decides; 0 waves = cell-column kernel).
template <typename T>
int run_wave_uniform(bool pack, int waves, unsigned w, const T* packed, T* unpacked, const T* refs, size_t ref_stride,
                     size_t n_blocks, void* stream)
{
    if (n_blocks == 0 || (pack && w == 0)) return FL_OK;
    if (!unpacked || (w != 0 && !packed)) return FL_ERR_NULL;
    if (misaligned(packed) || misaligned(unpacked)) return FL_ERR_ALIGN;
    WidthsArgs a;
    a.packed = reinterpret_cast<const char*>(packed);
    a.unpacked = reinterpret_cast<char*>(unpacked);
    a.widths = nullptr;
    a.offsets = nullptr;
    a.err_flag = nullptr;
    a.refs = refs;
    a.ref_stride = ref_stride;
    a.n_blocks = n_blocks;
    a.tiles_per_xcd = 0;
    a.window_shift = 63;
    a.uniform_width = w;
    a.bpw = uniform_blocks_per_wave(Elem<T>::BITS, pack, w, refs != nullptr);
    a.packed_bytes = 0;      // not read: uniform-width calls are validated here, on the host side
    a.prefetch = a.bpw > 1;
    a.linear_map = 0;
    a.nt_from = fl::nt_read_from(Elem<T>::BITS);
    const int pol = g_kernel_policy.load(std::memory_order_relaxed);       // A/B tools: 2 + ... + 65536 * blocks-per-wavefront (+ 2^24: prefetch)
    if ((pol & 0xff) == 2 && ((pol >> 16) & 0xff)) { a.bpw = (pol >> 16) & 0xff; a.prefetch = (pol >> 24) & 1; }
    if (a.prefetch && (WG / 64) * a.bpw * WaveBlock<T>::BLOCK_BYTES > 64u * 1024u) a.prefetch = 0;   // images would not fit a workgroup's LDS
    hipError_t e = widths_launcher<T>(pack)(a, waves, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? FL_OK : hip_fail(e);
}

template <typename T> int dev_pack(unsigned w, const T* in, T* out, size_t n, void* s)
{
    if (w > (unsigned)Elem<T>::BITS) return FL_ERR_WIDTH;
    if (const int waves = chosen_waves(Elem<T>::BITS, w, WAVE_PACK)) {
        if (n && !in) return FL_ERR_NULL;
        return run_wave_uniform<T>(true, waves, w, out, const_cast<T*>(in), nullptr, 0, n, s);
    }
    return run_stream<T>(pack_table_impl<T, PACK_PLAIN>().fn[w], in, out, nullptr, 0, n, true, w != 0, false, s);
}
template <typename T> int dev_unpack(unsigned w, const T* in, T* out, size_t n, void* s)
{
    if (w > (unsigned)Elem<T>::BITS) return FL_ERR_WIDTH;
    if (const int waves = chosen_waves(Elem<T>::BITS, w, WAVE_UNPACK))
        return run_wave_uniform<T>(false, waves, w, in, out, nullptr, 0, n, s);
    return run_stream<T>(unpack_table_impl<T, BODY_STORE>().fn[w], in, out, nullptr, 0, n, w != 0, true, false, s);
}
template <typename T>
int dev_for_pack(unsigned w, const T* in, const T* refs, size_t stride, T* out, size_t n, void* s)
{
    if (w > (unsigned)Elem<T>::BITS) return FL_ERR_WIDTH;
    if (const int waves = chosen_waves(Elem<T>::BITS, w, WAVE_FOR_PACK)) {
        if (n && (!in || !refs)) return FL_ERR_NULL;
        return run_wave_uniform<T>(true, waves, w, out, const_cast<T*>(in), refs, stride, n, s);
    }
    return run_stream<T>(pack_table_impl<T, PACK_FOR>().fn[w], in, out, refs, stride, n, true, w != 0, true, s);
}
template <typename T>
int dev_unfor_pack(unsigned w, const T* in, const T* refs, size_t stride, T* out, size_t n, void* s)
{
    if (w > (unsigned)Elem<T>::BITS) return FL_ERR_WIDTH;
    if (const int waves = chosen_waves(Elem<T>::BITS, w, WAVE_UNFOR_PACK)) {
        if (n && !refs) return FL_ERR_NULL;
        return run_wave_uniform<T>(false, waves, w, in, out, refs, stride, n, s);
    }
    return run_stream<T>(unpack_table_impl<T, BODY_ADD_REF>().fn[w], in, out, refs, stride, n, w != 0, true, true, s);
}
template <typename T>
int dev_undelta_pack(unsigned w, const T* in, const T* bases, T* out, size_t n, void* s)
{
    if (w > (unsigned)Elem<T>::BITS) return FL_ERR_WIDTH;
    if (n && misaligned(bases)) return FL_ERR_ALIGN;
    if (int waves = chosen_waves(Elem<T>::BITS, w, WAVE_UNDELTA_PACK)) {
        // a table entry 10 + k = the wave-per-block kernel with TWO blocks per wavefront in lockstep at k waves per SIMD (fl_dispatch.hpp);
        // the A/B tools force either form with the policy's blocks-per-wavefront field (2 + 256 * waves + 65536 * {1, 2})
        bool two = waves >= TWO_BLOCKS;
        if (two) waves -= TWO_BLOCKS;
        const int pol = g_kernel_policy.load(std::memory_order_relaxed);
        if ((pol & 0xff) == 2 && ((pol >> 16) & 0xff)) two = ((pol >> 16) & 0xff) == 2;
        const int rc = run_chain<T>(OP_UNDELTA_PACK, waves, w, in, bases, out, n, s, nullptr, nullptr, 0, nullptr, two);
        if (rc >= 0) return rc;                  // -1: no pipeline form of this op (cannot happen today): the cell-column kernel
    }
    return run_stream<T>(unpack_table_impl<T, BODY_UNDELTA>().fn[w], in, out, bases, 0, n, w != 0, true, true, s);
}
template <typename T>
int dev_undelta_pack_untranspose(unsigned w, const T* in, const T* bases, T* out, size_t n, void* s)
{
    if (w > (unsigned)Elem<T>::BITS) return FL_ERR_WIDTH;
    if (n && misaligned(bases)) return FL_ERR_ALIGN;
    if (const int waves = chosen_waves(Elem<T>::BITS, w, WAVE_UNDELTA_PACK_UNTRANSPOSE)) {
        const int rc = run_chain<T>(OP_UNDELTA_PACK_UNTRANSPOSE, waves, w, in, bases, out, n, s);
        if (rc >= 0) return rc;
    }
    return run_stream<T>(unpack_table_impl<T, BODY_UNDELTA_UNTRANSPOSE>().fn[w], in, out, bases, 0, n, w != 0, true, true, s);
}
template <typename T>
int dev_transpose_delta_pack(unsigned w, const T* in, const T* bases, T* out, size_t n, void* s)
{
    if (w > (unsigned)Elem<T>::BITS) return FL_ERR_WIDTH;
    if (n && misaligned(bases)) return FL_ERR_ALIGN;
    if (const int waves = chosen_waves(Elem<T>::BITS, w, WAVE_TRANSPOSE_DELTA_PACK)) {
        const int rc = run_chain<T>(OP_TRANSPOSE_DELTA_PACK, waves, w, in, bases, out, n, s);
        if (rc >= 0) return rc;
    }
    return run_stream<T>(pack_table_impl<T, PACK_TRANSPOSE_DELTA>().fn[w], in, out, bases, 0, n, true, w != 0, true, s);
}
template <typename T>
int dev_unpack_block_sums(unsigned w, const T* in, size_t n, uint64_t* sums, void* s)
{
    if (w > (unsigned)Elem<T>::BITS) return FL_ERR_WIDTH;
    if (n == 0) return FL_OK;
    if (!sums || (w != 0 && !in)) return FL_ERR_NULL;
    if (misaligned(in)) return FL_ERR_ALIGN;
    ReduceArgs a{reinterpret_cast<const u32x4*>(in), sums, nullptr, n, 0, 63};
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
    a.window_shift = 63;
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
    int waves = compare_launch_waves(Elem<T>::BITS, w);
    const int pol = g_kernel_policy.load(std::memory_order_relaxed);       // A/B tools: 2 + 256 * waves
    if ((pol & 0xff) == 2 && ((pol >> 8) & 0xff)) waves = (pol >> 8) & 0xff;
    hipError_t e = (a.is_eq ? compare_table_impl<T, true>() : compare_table_impl<T, false>()).fn[w](a, waves, static_cast<hipStream_t>(s));
    return e == hipSuccess ? FL_OK : hip_fail(e);
}
template <typename T>
int dev_block_min_max(const T* in, size_t n, T* mins, T* maxs, void* s)
{
    if (n == 0) return FL_OK;
    if (!in || !mins || !maxs) return FL_ERR_NULL;
    if (misaligned(in)) return FL_ERR_ALIGN;
    ReduceArgs a{reinterpret_cast<const u32x4*>(in), mins, maxs, n, 0, 63};
    hipError_t e = min_max_launcher<T>()(a, static_cast<hipStream_t>(s));
    return e == hipSuccess ? FL_OK : hip_fail(e);
}
template <typename T> int dev_delta(bool inverse, const T* in, const T* bases, T* out, size_t n, void* s)
{
    if (n && misaligned(bases)) return FL_ERR_ALIGN;
    if (const int waves = chosen_waves(Elem<T>::BITS, Elem<T>::BITS, inverse ? WAVE_UNDELTA : WAVE_DELTA)) {
        const int rc = run_chain<T>(inverse ? OP_UNDELTA : OP_DELTA, waves, Elem<T>::BITS, in, bases, out, n, s);
        if (rc >= 0) return rc;
    }
    return run_stream<T>(delta_launcher<T>(inverse), in, out, bases, 0, n, true, true, true, s);
}
template <typename T> int dev_transpose(bool inverse, const T* in, T* out, size_t n, void* s)
{
    if (const int waves = chosen_waves(Elem<T>::BITS, Elem<T>::BITS, inverse ? WAVE_UNTRANSPOSE : WAVE_TRANSPOSE)) {
        const int rc = run_chain<T>(inverse ? OP_UNTRANSPOSE : OP_TRANSPOSE, waves, Elem<T>::BITS, in, static_cast<const T*>(nullptr), out, n, s);
        if (rc >= 0) return rc;
    }
    return run_stream<T>(transpose_launcher<T>(inverse), in, out, nullptr, 0, n, true, true, false, s);
}
template <typename T>
int dev_unpack_single(unsigned w, const T* packed, size_t n_blocks, const uint64_t* idx, size_t n_idx,
                      T* out, uint32_t* err_flag, void* s)
{
    if (w > (unsigned)Elem<T>::BITS) return FL_ERR_WIDTH;
    if (n_idx == 0) return FL_OK;
    if (!idx || !out || (w != 0 && !packed)) return FL_ERR_NULL;
    SingleArgs a{packed, idx, out, err_flag, n_blocks, n_idx, w, nullptr, nullptr, 0};
    hipError_t e = unpack_single_launch<T>(a, static_cast<hipStream_t>(s));
    return e == hipSuccess ? FL_OK : hip_fail(e);
}

template <typename T>
int dev_unpack_single_widths(const uint8_t* widths, const uint64_t* offsets, const T* packed, size_t packed_bytes, size_t n_blocks,
                             const uint64_t* idx, size_t n_idx, T* out, uint32_t* err_flag, void* s)
{
    if (n_idx == 0) return FL_OK;
    static const T no_bytes[16 / sizeof(T)] __attribute__((aligned(16))) = {0};
    if (!packed && packed_bytes == 0) packed = no_bytes;      // a column of width-0 blocks has no packed bytes (every lookup is 0)
    if (!widths || !offsets || !idx || !out || !packed) return FL_ERR_NULL;
    SingleArgs a{packed, idx, out, err_flag, n_blocks, n_idx, 0, widths, offsets, packed_bytes};
    hipError_t e = unpack_single_launch<T>(a, static_cast<hipStream_t>(s));
    return e == hipSuccess ? FL_OK : hip_fail(e);
}

// ---------------------------------------------------------------------------
// Host tier: the trait methods' host slices, run through the same kernels.
//
// The reference is allocation-free (`#![no_std]`, lib.rs:3); so is this tier after its first call
// on a thread: every host thread keeps ONE cached context (HostCtx, thread_local) holding
//   * a private non-blocking stream (concurrent host threads do not serialise on the null stream),
//   * a pinned, device-mapped staging buffer and a device scratch buffer, both grown geometrically
//     and freed at thread exit or by fl_host_release().
// Small calls (one trait-method call = one block) are ZERO-COPY: the slices are copied into the
// pinned buffer and the kernel reads / writes that host memory directly over PCIe -- one launch, no DMA
// round trips, completion signalled through a word in pinned memory (HostCtx::wait_zero_copy).  Large calls
// stage through the device scratch buffer.
// ---------------------------------------------------------------------------
// the last thing queued behind a zero-copy call: one thread stores the call's sequence number into pinned host memory
__global__ void k_host_done(uint64_t* flag, uint64_t seq)
{
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// zero-copy calls whose completion word did not arrive within the spin's bound (each cost a 50 ms stall and a real synchronise):
// fl_internal_zero_copy_fallbacks() -- a 50-ms-per-call cliff must not be silent
std::atomic<uint64_t> g_zero_copy_fallbacks{0};

struct HostCtx {
    int device = -1;
    hipStream_t stream = nullptr;
    char* dev = nullptr;
    size_t dev_cap = 0;
    char* pin = nullptr;
    size_t pin_cap = 0;
    uint64_t* done = nullptr;      // pinned: the sequence number of the last finished zero-copy call (k_host_done)
    uint64_t seq = 0;
    unsigned since_sync = 0;

    // Best effort: runs from fl_host_release() and from the thread_local destructor, i.e. possibly while the process is
    // tearing down.  If the runtime no longer answers (hipGetDevice fails) or the context's device cannot be made current,
    // nothing is freed -- leaking at exit is harmless, calling into a torn-down runtime is not.  Long-lived worker
    // threads should call fl_host_release() themselves before they exit.
    void release()
    {
        if (device < 0) return;
        int cur = -1;
        bool usable = hipGetDevice(&cur) == hipSuccess;
        bool switched = false;
        if (usable && cur != device) usable = switched = hipSetDevice(device) == hipSuccess;
        if (usable) {
            if (stream) (void)hipStreamSynchronize(stream);
            if (stream) (void)hipStreamDestroy(stream);
            if (dev) (void)hipFree(dev);
            if (pin) (void)hipHostFree(pin);
            if (done) (void)hipHostFree(done);
            if (switched) (void)hipSetDevice(cur);
        }
        stream = nullptr; dev = nullptr; pin = nullptr; done = nullptr;
        dev_cap = pin_cap = 0;
        seq = 0; since_sync = 0;
        device = -1;
    }
    // Completion of everything queued on `stream` by a ZERO-COPY call (its results are in pinned host memory once the kernel has
    // retired).  hipStreamSynchronize costs ~9 of such a call's 13 us; a one-thread kernel queued behind the work that stores the
    // call's sequence number into pinned memory, and a host spin on that word, cost ~2.4 us less (tools/exp_host_sync.hip,
    // profiles/exp_host_sync_r04.txt: 13.1 -> 10.7 us).  The spin is bounded: if the number has not arrived after ~50 ms -- a kernel
    // that faulted never stores it -- or the marker cannot be launched, the stream is synchronised the ordinary way, which also
    // reports the error.  Every 4096th call synchronises for real so that the runtime retires its completion records.
    hipError_t wait_zero_copy()
    {
        if (!done || ++since_sync >= 4096) { since_sync = 0; return hipStreamSynchronize(stream); }
        const uint64_t want = ++seq;
        FL_LAUNCH(k_host_done, dim3(1), dim3(1), 0, stream, done, want);
        if (hipGetLastError() != hipSuccess) return hipStreamSynchronize(stream);
        std::chrono::steady_clock::time_point t0;
        for (unsigned spins = 0;; ++spins) {
            if (__atomic_load_n(done, __ATOMIC_ACQUIRE) == want) return hipSuccess;
            if ((spins & 0xffffu) == 0xffffu) {                   // every 65 536 polls (some tens of us): look at the clock
                const auto now = std::chrono::steady_clock::now();
                if (spins == 0xffffu) t0 = now;
                else if (now - t0 > std::chrono::milliseconds(50)) {                  // never seen in a healthy run: make it visible
                    g_zero_copy_fallbacks.fetch_add(1, std::memory_order_relaxed);
                    return hipStreamSynchronize(stream);
                }
            }
        }
    }
    // bind to the calling thread's current device
    hipError_t bind()
    {
        int cur = 0;
        hipError_t e = hipGetDevice(&cur);
        if (e != hipSuccess) return e;
        if (cur == device) return hipSuccess;
        release();
        e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
        if (e != hipSuccess) { stream = nullptr; return e; }
        device = cur;
        // fine-grained (coherent) and mapped, explicitly: the host must SEE the marker's system-scope store without a synchronising
        // call, which hipHostMallocDefault only implies
        if (hipHostMalloc(reinterpret_cast<void**>(&done), 64, hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess) *done = 0;
        else { done = nullptr; (void)hipGetLastError(); }          // no marker word: wait_zero_copy() synchronises the stream instead
        seq = 0;
        return hipSuccess;
    }
    static size_t grown(size_t need, size_t have) { return need > 2 * have ? need : 2 * have; }
    hipError_t need_pinned(size_t bytes)
    {
        if (bytes <= pin_cap) return hipSuccess;
        if (pin) {
            hipError_t es = hipStreamSynchronize(stream);          // a kernel may still be using the old buffer
            if (es != hipSuccess) return es;
            (void)hipHostFree(pin); pin = nullptr; pin_cap = 0;
        }
        const size_t cap = grown(bytes, pin_cap < 65536 ? 65536 : pin_cap);
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&pin), cap, hipHostMallocCoherent | hipHostMallocMapped);   // see `done`
        if (e == hipSuccess) pin_cap = cap; else pin = nullptr;
        return e;
    }
    hipError_t need_device(size_t bytes)
    {
        if (bytes <= dev_cap) return hipSuccess;
        if (dev) {
            hipError_t es = hipStreamSynchronize(stream);
            if (es != hipSuccess) return es;
            (void)hipFree(dev); dev = nullptr; dev_cap = 0;
        }
        const size_t cap = grown(bytes, dev_cap);
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&dev), cap);
        if (e == hipSuccess) dev_cap = cap; else dev = nullptr;
        return e;
    }
    ~HostCtx() { release(); }
};
thread_local HostCtx g_host;

constexpr size_t HOST_ZERO_COPY_LIMIT = 256 * 1024;   // bytes (in + aux + out) served straight from pinned host memory

#define FL_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return hip_fail(e_); } while (0)

inline size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }

// dev(in, aux, out, stream) launches the device-tier op on pointers the GPU can reach.
template <typename T, typename F>
int host_run(const T* in, size_t in_elems, const T* aux, size_t aux_elems, T* out, size_t out_elems, F&& dev)
{
    if ((in_elems && !in) || (out_elems && !out) || (aux_elems && !aux)) return FL_ERR_NULL;
    const size_t ib = in_elems * sizeof(T), ab = aux_elems * sizeof(T), ob = out_elems * sizeof(T);
    const size_t o_aux = pad256(ib), o_out = o_aux + pad256(ab), total = o_out + pad256(ob);
    HostCtx& c = g_host;
    FL_HIP(c.bind());
    fl::constructed_pair_this_thread() = false;             // the host tier's own staging buffers (a device-tier call may have left it set)
    if (total <= HOST_ZERO_COPY_LIMIT) {
        FL_HIP(c.need_pinned(total));
        if (ib) memcpy(c.pin, in, ib);
        if (ab) memcpy(c.pin + o_aux, aux, ab);
        int rc = dev(reinterpret_cast<const T*>(c.pin), reinterpret_cast<const T*>(c.pin + o_aux),
                     reinterpret_cast<T*>(c.pin + o_out), c.stream);
        if (rc != FL_OK) return rc;
        FL_HIP(c.wait_zero_copy());
        if (ob) memcpy(out, c.pin + o_out, ob);
        return FL_OK;
    }
    FL_HIP(c.need_device(total));
    if (ib) FL_HIP(hipMemcpyAsync(c.dev, in, ib, hipMemcpyHostToDevice, c.stream));
    if (ab) FL_HIP(hipMemcpyAsync(c.dev + o_aux, aux, ab, hipMemcpyHostToDevice, c.stream));
    int rc = dev(reinterpret_cast<const T*>(c.dev), reinterpret_cast<const T*>(c.dev + o_aux),
                 reinterpret_cast<T*>(c.dev + o_out), c.stream);
    if (rc != FL_OK) return rc;
    if (ob) FL_HIP(hipMemcpyAsync(out, c.dev + o_out, ob, hipMemcpyDeviceToHost, c.stream));
    FL_HIP(hipStreamSynchronize(c.stream));
    return FL_OK;
}

// unpack_single on host slices: only the indexed block travels (128*W bytes into the pinned buffer;
// the kernel then touches the one or two words bitpacking.rs:164-178 reads).
template <typename T>
int host_unpack_single(unsigned w, const T* pk, size_t n_blocks, uint64_t index, T* value)
{
    if (w > (unsigned)Elem<T>::BITS) return FL_ERR_WIDTH;
    if (!value) return FL_ERR_NULL;
    if (w == 0) { *value = 0; return FL_OK; }                 // bitpacking.rs:136-139 precedes the assert
    if (index >= (uint64_t)n_blocks * 1024) return FL_ERR_INDEX;   // bitpacking.rs:152
    if (!pk) return FL_ERR_NULL;
    const size_t pl = (size_t)1024 * w / Elem<T>::BITS, pb = pl * sizeof(T);
    const size_t o_idx = pad256(pb), o_val = o_idx + 256;
    HostCtx& c = g_host;
    FL_HIP(c.bind());
    FL_HIP(c.need_pinned(o_val + 256));
    memcpy(c.pin, pk + (index >> 10) * pl, pb);
    *reinterpret_cast<uint64_t*>(c.pin + o_idx) = index & 1023u;
    int rc = dev_unpack_single<T>(w, reinterpret_cast<const T*>(c.pin), 1, reinterpret_cast<const uint64_t*>(c.pin + o_idx), 1,
                                  reinterpret_cast<T*>(c.pin + o_val), nullptr, c.stream);
    if (rc != FL_OK) return rc;
    FL_HIP(c.wait_zero_copy());
    *value = *reinterpret_cast<const T*>(c.pin + o_val);
    return FL_OK;
}

template <typename T> size_t plen(unsigned w) { return (size_t)1024 * w / Elem<T>::BITS; }

}  // namespace

// ---------------------------------------------------------------------------
// mixed-width columns: device-resident widths[] / offsets[] (fl_widths.hpp)
// ---------------------------------------------------------------------------
struct fl_mixed_plan {
    unsigned type_bits = 0;
    size_t n_blocks = 0;
    uint64_t packed_bytes = 0;
    uint8_t* d_widths = nullptr;     // widths[n_blocks] in HBM
    uint64_t* d_offsets = nullptr;   // byte offset of every block in the packed column (exclusive prefix sum of 128*W)
};

namespace {

template <typename T>
int run_widths(bool pack, const uint8_t* widths, const uint64_t* offsets, const void* packed, size_t packed_bytes, void* unpacked,
               size_t n_blocks, uint32_t* err_flag, void* stream, const T* refs = nullptr, size_t ref_stride = 0, bool with_refs = false)
{
    if (n_blocks == 0) return FL_OK;
    if (with_refs && !refs) return FL_ERR_NULL;
    // a column whose blocks all have width 0 has no packed bytes at all: its packed pointer may be NULL (any block with a
    // width > 0 then fails the kernel's bounds check against packed_bytes == 0)
    static const char no_bytes[16] __attribute__((aligned(16))) = {0};
    if (!packed && packed_bytes == 0) packed = no_bytes;
    if (!widths || !offsets || !packed || !unpacked) return FL_ERR_NULL;
    if (misaligned(packed) || misaligned(unpacked)) return FL_ERR_ALIGN;
    WidthsArgs a;
    a.packed = static_cast<const char*>(packed);
    a.unpacked = static_cast<char*>(unpacked);
    a.widths = widths;
    a.offsets = offsets;
    a.err_flag = err_flag;
    a.refs = with_refs ? refs : nullptr;
    a.ref_stride = ref_stride;
    a.n_blocks = n_blocks;
    a.tiles_per_xcd = 0;
    a.window_shift = 63;
    a.uniform_width = 0;
    a.packed_bytes = packed_bytes;
    a.bpw = mixed_blocks_per_wave(Elem<T>::BITS, pack);
    a.prefetch = mixed_prefetch(Elem<T>::BITS);
    a.linear_map = 0;
    a.nt_from = 0;           // a mixed-width column always streams
    int waves = mixed_waves(Elem<T>::BITS, pack);
    const int pol = g_kernel_policy.load(std::memory_order_relaxed);       // A/B tools: 2 + 256*waves + 65536*blocks-per-wavefront (+ 2^24: prefetch)
    if ((pol & 0xff) == 2 && ((pol >> 8) & 0xff)) waves = (pol >> 8) & 0xff;
    if ((pol & 0xff) == 2 && ((pol >> 16) & 0xff)) { a.bpw = (pol >> 16) & 0xff; a.prefetch = (pol >> 24) & 1; }
    if (a.prefetch && (WG / 64) * a.bpw * WaveBlock<T>::BLOCK_BYTES > 64u * 1024u) a.prefetch = 0;   // images would not fit a workgroup's LDS
    hipError_t e = widths_launcher<T>(pack)(a, waves, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? FL_OK : hip_fail(e);
}

// Delta over a mixed-width column: the pipeline kernel with per-block widths[] / offsets[] on its packed side (fl_chain.hpp)
template <typename T>
int run_chain_widths(int op, const uint8_t* widths, const uint64_t* offsets, const T* in, const T* bases, T* out, size_t packed_bytes,
                     size_t n_blocks, uint32_t* err_flag, void* stream)
{
    if (n_blocks == 0) return FL_OK;
    if (!widths || !offsets) return FL_ERR_NULL;
    int waves = mixed_waves(Elem<T>::BITS, op == OP_TRANSPOSE_DELTA_PACK);
    if (sizeof(T) == 1) waves = 0;   // u8 runs the persistent pipelined kernels: 0 = their own grid
    const int pol = g_kernel_policy.load(std::memory_order_relaxed);       // A/B tools: 2 + 256 * waves
    if ((pol & 0xff) == 2 && ((pol >> 8) & 0xff)) waves = (pol >> 8) & 0xff;
    const int rc = run_chain<T>(op, waves, 0, in, bases, out, n_blocks, stream, widths, offsets, packed_bytes, err_flag);
    return rc >= 0 ? rc : hip_fail(hipErrorInvalidDeviceFunction);
}

template <typename T> int dev_for_widths(const T* mins, const T* maxs, size_t n, uint8_t* widths, void* s)
{
    if (n == 0) return FL_OK;
    if (!mins || !maxs || !widths) return FL_ERR_NULL;
    hipError_t e = launch_for_widths<T>(mins, maxs, n, widths, static_cast<hipStream_t>(s));
    return e == hipSuccess ? FL_OK : hip_fail(e);
}

// many small arrays, device arrays of pointers (fl_batch.hpp)
template <typename T>
int run_batch(bool pack, const void* const* packed, void* const* unpacked, const uint8_t* widths, const void* refs, bool with_refs,
              const uint32_t* n_blocks, size_t n_arrays, uint32_t max_blocks, uint32_t* err_flag, void* stream)
{
    if (n_arrays == 0 || max_blocks == 0) return FL_OK;
    if (!packed || !unpacked || !widths || !n_blocks || (with_refs && !refs)) return FL_ERR_NULL;
    if (max_blocks > BATCH_MAX_BLOCKS) return FL_ERR_INDEX;       // 2^30 blocks = 2^40 values in ONE array: a bound nobody means
    BatchArgs b;
    b.packed = reinterpret_cast<const char* const*>(packed);
    b.unpacked = reinterpret_cast<char* const*>(unpacked);
    b.widths = widths;
    b.n_blocks = n_blocks;
    b.err_flag = err_flag;
    b.refs = with_refs ? refs : nullptr;
    b.bases = nullptr;
    b.n_arrays = n_arrays;
    b.tiles_per_xcd = 0;
    b.window_shift = 63;
    b.tiles_per_array = 0;
    b.max_blocks = max_blocks;
    b.bpw = batch_blocks_per_wave(Elem<T>::BITS, pack);
    b.prefetch = b.bpw > 1;
    int waves = batch_waves(Elem<T>::BITS, pack);
    const int pol = g_kernel_policy.load(std::memory_order_relaxed);       // A/B tools: as for the mixed-width kernels (run_widths)
    if ((pol & 0xff) == 2 && ((pol >> 8) & 0xff)) waves = (pol >> 8) & 0xff;
    if ((pol & 0xff) == 2 && ((pol >> 16) & 0xff)) { b.bpw = (pol >> 16) & 0xff; b.prefetch = (pol >> 24) & 1; }
    hipError_t e = batch_launcher<T>(pack)(b, max_blocks, waves, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? FL_OK : hip_fail(e);
}

// Delta over many small arrays (fl_batch.hpp: k_batch_chain)
template <typename T>
int run_batch_chain(int op, const void* const* packed, const void* const* bases, void* const* unpacked, const uint8_t* widths,
                    const uint32_t* n_blocks, size_t n_arrays, uint32_t max_blocks, uint32_t* err_flag, void* stream)
{
    if (n_arrays == 0 || max_blocks == 0) return FL_OK;
    if (!packed || !bases || !unpacked || !widths || !n_blocks) return FL_ERR_NULL;
    if (max_blocks > BATCH_MAX_BLOCKS) return FL_ERR_INDEX;
    const batch_launch_t fn = batch_chain_launcher<T>(op);
    if (!fn) return hip_fail(hipErrorInvalidDeviceFunction);
    BatchArgs b;
    b.packed = reinterpret_cast<const char* const*>(packed);
    b.unpacked = reinterpret_cast<char* const*>(unpacked);
    b.widths = widths;
    b.n_blocks = n_blocks;
    b.err_flag = err_flag;
    b.refs = nullptr;
    b.bases = reinterpret_cast<const char* const*>(bases);
    b.n_arrays = n_arrays;
    b.tiles_per_xcd = 0;
    b.window_shift = 63;
    b.tiles_per_array = 0;
    b.max_blocks = max_blocks;
    b.bpw = 1;
    b.prefetch = 0;
    int waves = mixed_waves(Elem<T>::BITS, op == OP_TRANSPOSE_DELTA_PACK);
    const int pol = g_kernel_policy.load(std::memory_order_relaxed);       // A/B tools: 2 + 256 * waves
    if ((pol & 0xff) == 2 && ((pol >> 8) & 0xff)) waves = (pol >> 8) & 0xff;
    hipError_t e = fn(b, max_blocks, waves, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? FL_OK : hip_fail(e);
}

template <typename T>
int run_mixed(bool pack, const fl_mixed_plan* p, const void* packed, void* unpacked, void* stream)
{
    if (!p) return FL_ERR_NULL;
    if (p->type_bits != (unsigned)Elem<T>::BITS) return FL_ERR_WIDTH;
    if (p->n_blocks == 0) return FL_OK;
    if (!unpacked || (p->packed_bytes && !packed)) return FL_ERR_NULL;
    // widths were validated at plan creation; an all-zero-width column has no packed bytes at all (run_widths accepts NULL then)
    return run_widths<T>(pack, p->d_widths, p->d_offsets, p->packed_bytes ? packed : nullptr, p->packed_bytes, unpacked, p->n_blocks, nullptr, stream);
}

}  // namespace

extern "C" {

int fl_widths_to_offsets(unsigned type_bits, const uint8_t* widths, size_t n_blocks, uint64_t* offsets,
                         uint64_t* total_bytes, uint32_t* err_flag, void* stream)
{
    if (type_bits != 8 && type_bits != 16 && type_bits != 32 && type_bits != 64) return FL_ERR_WIDTH;
    if (n_blocks && (!widths || !offsets)) return FL_ERR_NULL;
    FL_DEVICE_TIER(stream, widths, offsets, total_bytes, err_flag);
    ScanArgs a{widths, offsets, total_bytes, err_flag, n_blocks, type_bits};
    hipError_t e = launch_widths_to_offsets(a, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? FL_OK : hip_fail(e);
}

int fl_mixed_plan_create(unsigned type_bits, const uint8_t* widths, size_t n_blocks, fl_mixed_plan** plan)
{
    if (!plan || (n_blocks && !widths)) return FL_ERR_NULL;
    *plan = nullptr;
    if (type_bits != 8 && type_bits != 16 && type_bits != 32 && type_bits != 64) return FL_ERR_WIDTH;
    for (size_t b = 0; b < n_blocks; ++b)
        if (widths[b] > type_bits) return FL_ERR_WIDTH;   // bitpacking.rs:93 unreachable!()
    fl_mixed_plan* p = new (std::nothrow) fl_mixed_plan;
    if (!p) { g_last_hip_error = (int)hipErrorOutOfMemory; return FL_ERR_HIP; }
    p->type_bits = type_bits;
    p->n_blocks = n_blocks;
    if (n_blocks) {
        uint64_t* d_total = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p->d_widths), n_blocks);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_offsets), (n_blocks + 1) * sizeof(uint64_t));
        if (e == hipSuccess) {
            d_total = p->d_offsets + n_blocks;             // the total rides behind the offsets: one allocation less
            e = hipMemcpy(p->d_widths, widths, n_blocks, hipMemcpyHostToDevice);
        }
        if (e == hipSuccess) {
            ScanArgs a{p->d_widths, p->d_offsets, d_total, nullptr, n_blocks, type_bits};
            e = launch_widths_to_offsets(a, nullptr);
        }
        if (e == hipSuccess) e = hipMemcpy(&p->packed_bytes, d_total, sizeof(uint64_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            fl_mixed_plan_destroy(p);
            return hip_fail(e);
        }
    }
    *plan = p;
    return FL_OK;
}

int fl_fill_random(void* dst, size_t n_bytes, uint64_t seed, void* stream)
{
    if (n_bytes == 0) return FL_OK;
    if (!dst) return FL_ERR_NULL;
    if ((reinterpret_cast<uintptr_t>(dst) & 7u) || (n_bytes & 7u)) return FL_ERR_ALIGN;
    FL_DEVICE_TIER(stream, dst);
    hipError_t e = launch_fill_random(static_cast<uint64_t*>(dst), n_bytes / 8, seed, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? FL_OK : hip_fail(e);
}

// ---- the bare stream (fl_stream.hpp) and the launch shape the library gives an op -----------------------------------------------
int fl_internal_bare_stream(const void* in, size_t in_unit, const void* aux, size_t aux_unit, void* out, size_t out_unit, size_t n_units,
                            int nt_loads, int waves, int window_log2_units, void* stream)
{
    if (n_units == 0) return FL_OK;
    if ((in_unit && !in) || (aux_unit && !aux) || !out) return FL_ERR_NULL;
    if (misaligned(in) || misaligned(aux) || misaligned(out)) return FL_ERR_ALIGN;
    if (in_unit > BARE_MAX_UNIT || out_unit > BARE_MAX_UNIT || aux_unit > 1024 || ((in_unit | out_unit | aux_unit) & 15u)) return FL_ERR_INDEX;
    FL_DEVICE_TIER(stream, in, aux, out);
    BareArgs a{static_cast<const char*>(in), static_cast<const char*>(aux), static_cast<char*>(out), n_units, 0,
               (unsigned)in_unit, (unsigned)aux_unit, (unsigned)out_unit, 63u};
    hipError_t e = launch_bare_stream(a, nt_loads != 0, waves, window_log2_units, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? FL_OK : hip_fail(e);
}

// op: 0 unpack / unfor_pack, 1 pack / for_pack, 2 undelta_pack, 3 unpack over a mixed-width column (width = the column's mean width
// times 2, so that 16.5 can be said).  The shape a bare stream must have to shadow that call: bytes per block on either side, the
// cache policy of the loads, waves per SIMD and tile-map window the library's own kernel for that (T, W) runs with.
int fl_internal_bare_stream_shape(int op, unsigned type_bits, unsigned width, size_t* in_unit, size_t* aux_unit, size_t* out_unit,
                                  int* nt_loads, int* waves, int* window_log2_units, unsigned* blocks_per_unit)
{
    if (type_bits != 8 && type_bits != 16 && type_bits != 32 && type_bits != 64) return FL_ERR_INDEX;
    if (op < 0 || op > 3) return FL_ERR_INDEX;
    if (width > (op == 3 ? 2 * type_bits : type_bits)) return FL_ERR_WIDTH;
    if (!in_unit || !aux_unit || !out_unit || !nt_loads || !waves || !window_log2_units || !blocks_per_unit) return FL_ERR_NULL;
    // a wavefront's unit is at least 4 KiB of unpacked values: 4 consecutive u8 blocks, 2 u16 blocks -- the library's own kernels never
    // give a wavefront a single 1- or 2-KiB block either (8 blocks per wavefront in the cell-column kernels, 2-4 in flight in the others),
    // and a stream that did would measure the starving wavefront, not the memory
    const unsigned k = type_bits == 8 ? 4u : type_bits == 16 ? 2u : 1u;
    const size_t packed = (op == 3 ? 64u * width : 128u * width) * k, unpacked = 128u * type_bits * k;
    *blocks_per_unit = k;
    *in_unit = op == 1 ? unpacked : packed;
    *out_unit = op == 1 ? packed : unpacked;
    *aux_unit = op == 2 ? 128u * k : 0;
    const WaveOp wop = op == 1 ? WAVE_PACK : op == 2 ? WAVE_UNDELTA_PACK : WAVE_UNPACK;
    int w = op == 3 ? mixed_waves(type_bits, false) : chosen_waves(type_bits, width, wop);
    if (w >= TWO_BLOCKS) w -= TWO_BLOCKS;                           // two blocks per wavefront: same bytes per launch, the occupancy is what the table says
    if (w == 0) w = 8;       // a cell-column kernel gives a wavefront 8 blocks at 2-3 waves per SIMD: the one-unit-per-wavefront stream needs every slot to keep as many bytes in flight
    *waves = w < 3 ? 3 : w;
    *nt_loads = op == 1 || op == 3 || width >= fl::nt_read_from(type_bits);       // fl_widths.hpp: RD_AUTO; pack reads non-temporally
    *window_log2_units = window_log2_blocks(op == 1 ? WIN_PACK : op == 2 ? WIN_UNDELTA_PACK : WIN_UNPACK, type_bits);
    return FL_OK;
}

// ---- the class of every piece of a stretch of device memory, by measurement (fastlanes_amd_internal.h: fl_internal_probe_memory_classes;
// FL_LAYOUT_INTERLEAVED below) ---------------------------------------------------------------------------------------------------------
namespace {
// classes[g] = 0, 1, 2 (or -1: no clean answer) for the n pieces of `piece` bytes at `base`: unpack_compare u32 W=20 reads the start of a
// representative piece and writes its 1/20 mask into piece g (at piece - min(piece / 2, 1 GiB)); the slow ones are of the
// representative's class (7.0 TB/s across classes, 6.05 TB/s inside one: profiles/exp_region_map_r03.txt, profiles/r06_vmm_placement.txt)
int probe_classes(char* base, size_t n, size_t piece, int* classes, hipStream_t s)
{
    constexpr unsigned PROBE_WIDTH = 20;
    const size_t probe_blocks = std::min<size_t>(2000000, piece / (128 * PROBE_WIDTH));
    const size_t mask_off = piece - std::min<size_t>(piece / 2, (size_t)1 << 30);
    for (size_t g = 0; g < n; ++g) classes[g] = -1;
    if (n == 0 || probe_blocks == 0) return FL_OK;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    hipError_t e = hipEventCreate(&t0);
    if (e == hipSuccess) e = hipEventCreate(&t1);
    int rc = e == hipSuccess ? FL_OK : hip_fail(e);
    // milliseconds of the probe reading piece gi and writing into piece gm: median of 3 after one untimed launch
    auto probe_ms = [&](size_t gi, size_t gm, float& ms) -> int {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(base + gi * piece);
        uint32_t* mask = reinterpret_cast<uint32_t*>(base + gm * piece + mask_off);
        float t[3] = {0.f, 0.f, 0.f};
        for (int i = -1; i < 3; ++i) {
            hipError_t h = hipEventRecord(t0, s);
            if (h != hipSuccess) return hip_fail(h);
            // under the whole-column tile map the class map was characterised with (a windowed read stream interferes less with the
            // thin write stream, which is the point of the window and blunts the probe): an override for THIS THREAD's launches only --
            // concurrent calls of other threads keep their own tile maps, and a concurrent fl_internal_set_kernel_policy is untouched
            const int saved = fl::window_override_this_thread();
            fl::window_override_this_thread() = fl::WINDOW_WHOLE;
            const int r = fl_u32_unpack_compare(PROBE_WIDTH, src, FL_CMP_LT, 1u << (PROBE_WIDTH - 1), probe_blocks, mask, s);
            fl::window_override_this_thread() = saved;
            if (r != FL_OK) return r;
            h = hipEventRecord(t1, s);
            if (h == hipSuccess) h = hipEventSynchronize(t1);
            float x = 0.f;
            if (h == hipSuccess) h = hipEventElapsedTime(&x, t0, t1);
            if (h != hipSuccess) return hip_fail(h);
            if (i >= 0) t[i] = x;
        }
        std::sort(t, t + 3);
        ms = t[1];
        return FL_OK;
    };
    float threshold = 0.f;                                   // between the two levels, from the first representative
    std::vector<float> ms(n);
    for (int c = 0; c < 3 && rc == FL_OK; ++c) {
        size_t rep = 0;
        while (rep < n && classes[rep] != -1) ++rep;
        if (rep == n) break;
        classes[rep] = c;
        rc = fl_fill_random(base + rep * piece, probe_blocks * 128 * PROBE_WIDTH, 17 + rep, s);   // full-entropy probe input
        float slowest = 0.f, fastest = 1e30f;
        size_t others = 0;
        for (size_t g = 0; g < n && rc == FL_OK; ++g) {
            if (classes[g] != -1) continue;
            rc = probe_ms(rep, g, ms[g]);
            slowest = std::max(slowest, ms[g]);
            fastest = std::min(fastest, ms[g]);
            ++others;
        }
        if (rc != FL_OK || others == 0) break;
        if (threshold == 0.f) {
            if (slowest - fastest <= 0.05f * slowest) break;     // one level only: "all of my class" and "none of it" look the same
            threshold = 0.5f * (slowest + fastest);
        }
        for (size_t g = 0; g < n; ++g)
            if (classes[g] == -1 && ms[g] > threshold) classes[g] = c;
    }
    if (t0) (void)hipEventDestroy(t0);
    if (t1) (void)hipEventDestroy(t1);
    return rc;
}
}  // namespace

int fl_internal_probe_memory_classes(void* slab, size_t slab_bytes, int* classes, void* stream)
{
    const size_t n_granules = slab_bytes / FL_INTERNAL_GRANULE_BYTES;
    if (n_granules == 0) return FL_OK;
    if (!slab || !classes) return FL_ERR_NULL;
    if (misaligned(slab)) return FL_ERR_ALIGN;
    return probe_classes(static_cast<char*>(slab), n_granules, FL_INTERNAL_GRANULE_BYTES, classes, static_cast<hipStream_t>(stream));
}

// ---- fl_column_pair_alloc / _free: the OPTIONAL allocation helper of fastlanes_amd.h ------------------------------------------------
namespace {
constexpr size_t PAIR_ALIGN = 256, PAIR_ZONE = (size_t)64 << 30, PAIR_MIB = (size_t)1 << 20, PAIR_GIB = (size_t)1 << 30;
constexpr size_t PAIR_INTERLEAVED_MIN = 8 * PAIR_GIB;       // below that a pair is a handful of chunks: nothing to arrange
inline size_t pair_pad(size_t b) { return (b + PAIR_ALIGN - 1) & ~(PAIR_ALIGN - 1); }

void live_pair_remove(const char* va);                   // the launchers' registry of live constructed pairs, below

// A bounded cache of 1-GiB physical chunks (fl_internal_pair_chunk_cache: OFF unless a tool asks for it).  hipMemCreate costs ~30 ms per
// GiB, so a constructed pair costs seconds, nearly all of it creating chunks that are released again a moment later; a sweep that builds a
// pair per row keeps them instead.  Only the handles are kept -- every pool is classified afresh (2.4 ms per chunk), class labels do not
// carry over from one probe to the next.
constexpr int CACHE_DEVICES = 16;
std::vector<hipMemGenericAllocationHandle_t> g_chunk_cache[CACHE_DEVICES];
size_t g_chunk_cache_limit = 0;
std::atomic_flag g_chunk_cache_lock = ATOMIC_FLAG_INIT;
void recycle_chunk(hipMemGenericAllocationHandle_t h, int dev)
{
    bool kept = false;
    if (dev >= 0 && dev < CACHE_DEVICES) {
        while (g_chunk_cache_lock.test_and_set(std::memory_order_acquire)) {}
        if (g_chunk_cache[dev].size() < g_chunk_cache_limit) { g_chunk_cache[dev].push_back(h); kept = true; }
        g_chunk_cache_lock.clear(std::memory_order_release);
    }
    if (!kept) (void)hipMemRelease(h);
}
bool cached_chunk(int dev, hipMemGenericAllocationHandle_t& h)
{
    bool got = false;
    if (dev >= 0 && dev < CACHE_DEVICES) {
        while (g_chunk_cache_lock.test_and_set(std::memory_order_acquire)) {}
        if (!g_chunk_cache[dev].empty()) { h = g_chunk_cache[dev].back(); g_chunk_cache[dev].pop_back(); got = true; }
        g_chunk_cache_lock.clear(std::memory_order_release);
    }
    return got;
}

struct ColumnPair {
    void* bufs[3] = {nullptr, nullptr, nullptr};       // separate: in, aux, out; zoned: the slab only
    void *in = nullptr, *aux = nullptr, *out = nullptr;
    // FL_LAYOUT_INTERLEAVED: physical chunks (hipMemCreate) mapped into one reserved address range
    std::vector<hipMemGenericAllocationHandle_t> chunks;
    char* va = nullptr;
    size_t va_bytes = 0, chunk_bytes = 0, n_mapped = 0;
    int dev = -1;                                        // the device the chunks belong to
    char class_map[96] = {0};                            // 'A' 'B' 'C' '?' per mapped chunk (first 95), input first
    void release()
    {
        for (void*& b : bufs) {
            if (b) (void)hipFree(b);
            b = nullptr;
        }
        for (size_t i = 0; i < n_mapped; ++i) (void)hipMemUnmap(va + i * chunk_bytes, chunk_bytes);
        n_mapped = 0;
        for (auto h : chunks) recycle_chunk(h, dev);
        chunks.clear();
        if (va) live_pair_remove(va);
        if (va) (void)hipMemAddressFree(va, va_bytes);
        va = nullptr;
    }
};

// Address ranges for FL_LAYOUT_INTERLEAVED.  On this ROCm (7.2) an address range that held a mapping, was unmapped and is mapped AGAIN --
// even after hipMemAddressFree + hipMemAddressReserve -- keeps translating to the chunks it held FIRST (tools/exp_vmm remap,
// profiles/r06_vmm_placement.txt): kernels would silently read and write memory that is no longer ours.  So no range is ever used twice
// within a process: ranges are asked for at monotonically growing addresses of a private stretch of the address space (16 .. 112 TiB; a hint that collides with something mapped there just yields another address: enough for several hundred pairs -- a pair uses its own size plus its pool's, once; then hipErrorOutOfMemory), and
// whatever the runtime returns is checked against every range this library used before.
std::atomic<uintptr_t> g_va_next{(uintptr_t)0x100000000000ull};
constexpr uintptr_t VA_ARENA_END = (uintptr_t)0x700000000000ull;
std::atomic_flag g_va_lock = ATOMIC_FLAG_INIT;
std::vector<std::pair<uintptr_t, uintptr_t>> g_va_used;

hipError_t reserve_fresh_range(size_t bytes, char** out)
{
    std::vector<void*> rejected;
    hipError_t e = hipErrorOutOfMemory;
    *out = nullptr;
    for (int attempt = 0; attempt < 6 && !*out; ++attempt) {
        const uintptr_t span = ((bytes + PAIR_GIB - 1) & ~(PAIR_GIB - 1)) + PAIR_GIB;
        const uintptr_t hint = g_va_next.fetch_add(span, std::memory_order_relaxed);
        if (hint + span > VA_ARENA_END) { e = hipErrorOutOfMemory; break; }
        void* p = nullptr;
        e = hipMemAddressReserve(&p, bytes, 2 * PAIR_MIB, reinterpret_cast<void*>(hint), 0);
        if (e != hipSuccess) { (void)hipGetLastError(); continue; }
        const uintptr_t lo = reinterpret_cast<uintptr_t>(p), hi = lo + bytes;
        while (g_va_lock.test_and_set(std::memory_order_acquire)) {}
        bool used = false;
        for (const auto& r : g_va_used) used = used || (lo < r.second && r.first < hi);
        if (!used) g_va_used.emplace_back(lo, hi);
        g_va_lock.clear(std::memory_order_release);
        if (used) { rejected.push_back(p); e = hipErrorOutOfMemory; }    // held until the end so that the runtime offers another one
        else *out = static_cast<char*>(p);
    }
    for (void* p : rejected) (void)hipMemAddressFree(p, bytes);
    return *out ? hipSuccess : e;
}

// which chunks of a classified pool form the pair (profiles/r06_vmm_placement.txt, r06_vmm/vmm_ratio_*.txt, vmm_pos_*.txt; fractions of 8 TB/s):
//  * the input (+ aux) inside ONE class (reads spread over classes under the whole-column tile map: 0.80 where one class gives 0.84-0.86);
//  * the output over `out_classes` classes: the OTHER TWO for a write-dominated pair (u32 W=7 unpack: out BC 0.864-0.866, out ABC 0.860,
//    out AB 0.855, out B 0.80, out A 0.78), ALL THREE otherwise (pack u32 W=7: out ABC 0.859, out AB 0.85, out BC 0.84, out B 0.83;
//    transpose: ABC 0.880, BC 0.870-0.879; unpack u32 W=20: equal);
//  * arranged for the WRITE POSITIONS, not in fixed runs: under the whole-column tile map XCD x walks the x-th eighth of the output, all
//    eight at the same pace, so at progress t the eight positions are the chunks floor((x + t) * n_out / 8).  What the memory wants is
//    those eight spread evenly over the classes AT EVERY t.  A fixed run length resonates with the eighth's size for some column lengths
//    (runs of 2 GiB at 8 M blocks: seven of the eight positions in one class, 0.840 where a balanced arrangement gives 0.864; runs of 1 GiB
//    with three classes at 6.5 M blocks: 0.829 / 0.859), and so does any closed formula once a chunk straddles two eighths (n_out = 20: the
//    "k-th chunk of eighth x takes letter x + k" rule puts all eight positions into one class at t = 0 -- unpack u16 W=3 0.852 where runs of
//    two had 0.868).  So the arrangement is SEARCHED: cost = mean over 64 values of t of the eight positions' cubed class
//    counts (an even spread is cheapest), plus a large penalty per chunk a class does not have and a fee per chunk of a
//    class outside the rotation (the input's own class: in A | out A and B alternating is 0.855, out B alone 0.80; unclassified chunks
//    last); start = letters by the position of a chunk's centre, then single-chunk relabelling until nothing improves (n_out * 4 * 64 * 8
//    operations per sweep: microseconds).  Creation order (short class runs by nature: 0.854-0.861) when no class can hold the input.
void choose_chunks(const std::vector<int>& cls, size_t n_in, size_t n_out, int out_classes, std::vector<int>& order, size_t* kept_out = nullptr)
{
    if (kept_out) *kept_out = 0;
    const size_t n = cls.size();
    order.clear();
    if (out_classes != 2) out_classes = 3;
    std::vector<int> by[4];                                    // 0..2 = classes, 3 = unclassified
    for (size_t g = 0; g < n; ++g) by[cls[g] < 0 || cls[g] > 2 ? 3 : cls[g]].push_back((int)g);
    if (n_out == 0 || n_in + n_out > n) {                      // nothing to arrange / the pool is too small: as created, as far as it goes
        for (size_t g = 0; g < n_in + n_out && g < n; ++g) order.push_back((int)g);
        return;
    }
    constexpr int TS = 64;                                     // samples of the progress t
    // pos[t][x] = the chunk under XCD x's write position at progress (t + 0.5) / TS
    std::vector<unsigned> pos((size_t)TS * 8);
    for (int t = 0; t < TS; ++t)
        for (int x = 0; x < 8; ++x) {
            const double p = (x + (t + 0.5) / TS) * (double)n_out / 8.0;
            pos[(size_t)t * 8 + x] = (unsigned)std::min<double>((double)n_out - 1, p);
        }
    struct Plan { std::vector<int> label; double cost; size_t kept; };
    auto plan = [&](int c) {
        Plan P;
        size_t avail[4] = {by[0].size(), by[1].size(), by[2].size(), by[3].size()};
        avail[c] -= n_in;
        // cost = the mean over t of sum_k count_k^3 (8 positions: 4 + 4 + 0 costs 128, 8/3 each 57, all in one class 512) + a fee per
        // FRACTION of the output taken from outside the rotation.  The cube and 600 for the input's own class rank the layouts as measured
        // (unpack u32 W=7, input in A): out BC 128 (0.865) < ABC 257 (0.860) < AB 428 (0.855) < B alone 512 (0.80) < A alone 1112 (0.78);
        // with squares no fee ranks "in B | out four fifths A" behind "in A | out A and B alternating" AND keeps BC ahead of ABC.  Below
        // ~300 the search sprinkles chunks of the input's class into a balanced pool's output (one such chunk gains 288 / n_out by taking a
        // position out of a 4 + 4 split): measured -0.4 % at 150, -0.0 ... -0.4 % at 300 against 1000 (profiles/r06_exp_own_class_fee.txt);
        // above 768 "B alone" would beat "A and B alternating" where a class is missing.
        static const double own_class_fee = [] { const char* e = getenv("FL_INTERNAL_OWN_CLASS_FEE"); return e ? atof(e) : 600.0; }();   // A/B tools
        double fee[4] = {0, 0, 0, own_class_fee + 50.0};
        bool in_rotation[3], plentiful = true;
        for (int k = 0; k < 3; ++k) {
            in_rotation[k] = out_classes == 3 || k != c;
            if (in_rotation[k] && avail[k] * (size_t)out_classes < n_out + (size_t)out_classes - 1) plentiful = false;
        }
        // ... where the rotation's classes cannot cover the output evenly the input's class has to help, and half the fee ranks "the scarce
        // class + the input's class + the plentiful one, evenly" ahead of "two thirds in the plentiful class" (AB 0.855 against B alone 0.80)
        for (int k = 0; k < 3; ++k) fee[k] = in_rotation[k] ? 0.0 : plentiful ? own_class_fee : 0.5 * own_class_fee;
        const int S[3] = {(c + 1) % 3, (c + 2) % 3, c};
        P.label.assign(n_out, 0);
        for (size_t j = 0; j < n_out; ++j) {                   // start: by the position of the chunk's centre
            const double at = (j + 0.5) * 8.0 / (double)n_out;
            const size_t x = (size_t)at, k = (size_t)((at - (double)x) * (double)n_out / 8.0);
            P.label[j] = S[(x + k) % (size_t)out_classes];
        }
        auto cost_of = [&](const std::vector<int>& lab) {
            double cost = 0.0;
            size_t used[4] = {0, 0, 0, 0};
            for (int k : lab) { ++used[k]; cost += fee[k] / (double)n_out; }
            for (int k = 0; k < 4; ++k) if (used[k] > avail[k]) cost += 10000.0 * (double)(used[k] - avail[k]);
            for (int t = 0; t < TS; ++t) {
                double cnt[4] = {0, 0, 0, 0};
                for (int x = 0; x < 8; ++x) cnt[lab[pos[(size_t)t * 8 + x]]] += 1.0;
                for (int k = 0; k < 4; ++k) cost += cnt[k] * cnt[k] * cnt[k] / TS;
            }
            return cost;
        };
        P.cost = cost_of(P.label);
        for (int sweep = 0; sweep < 12; ++sweep) {
            bool improved = false;
            for (size_t j = 0; j < n_out; ++j) {
                const int was = P.label[j];
                int best = was;
                for (int k = 0; k < 4; ++k) {
                    if (k == was) continue;
                    P.label[j] = k;
                    const double cst = cost_of(P.label);
                    if (cst < P.cost - 1e-9) { P.cost = cst; best = k; }
                }
                P.label[j] = best;
                improved = improved || best != was;
            }
            if (!improved) break;
        }
        // "kept" = chunks of rotation classes, less what the class shares are out of balance by (the pool-growth criterion)
        size_t used[4] = {0, 0, 0, 0};
        for (int k : P.label) ++used[k];
        double off = 0.0;
        size_t outside = used[3];
        for (int k = 0; k < 3; ++k) {
            if (in_rotation[k]) off += std::max(0.0, std::fabs((double)used[k] - (double)n_out / out_classes) - 1.0);
            else outside += used[k];
        }
        const double kept = (double)n_out - (double)outside - off;
        P.kept = kept > 0.0 ? (size_t)kept : 0;
        for (int k = 0; k < 4; ++k) if (used[k] > avail[k]) P.kept = 0;      // cannot even be filled
        return P;
    };
    int best_c = -1;
    Plan best;
    for (int c = 0; c < 3; ++c) {
        if (by[c].size() < n_in) continue;
        Plan P = plan(c);
        if (best_c < 0 || P.cost < best.cost - 1e-9 || (std::fabs(P.cost - best.cost) <= 1e-9 && by[c].size() > by[best_c].size())) { best = std::move(P); best_c = c; }
    }
    if (best_c < 0) {                                           // no class can hold the input: as created
        for (size_t g = 0; g < n_in + n_out && g < n; ++g) order.push_back((int)g);
        return;
    }
    if (kept_out) *kept_out = best.kept;
    size_t next[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < n_in; ++i) order.push_back(by[best_c][next[best_c]++]);
    for (size_t j = 0; j < n_out; ++j) {
        int k = best.label[j];
        if (next[k] >= by[k].size()) {                          // (only if the penalty lost against the imbalance: any chunk that is left)
            k = -1;
            for (int q = 0; q < 4; ++q) if (next[q] < by[q].size() && (k < 0 || by[q].size() - next[q] > by[k].size() - next[k])) k = q;
            if (k < 0) break;
        }
        order.push_back(by[k][next[k]++]);
    }
}

// Live FL_LAYOUT_INTERLEAVED pairs, for the launchers: a call whose buffers lie inside ONE constructed pair runs under the whole-column tile
// map whatever fl_window_table.inc says -- the table's windows (pack, the transposes, delta ...) are what plain allocations want (a window
// keeps the eight XCDs' reads inside one class of memory), while a constructed pair already has its input inside one class and wants the
// eight write positions spread over the output's rotation: w=31 never loses there and gains 1-3 % on every row the table windows
// (profiles/r06_window_matrix_constructed.txt; pack u32 W=7 0.821 -> 0.836, then 0.859 with the three-class output rotation).
constexpr int LIVE_PAIRS = 32;
std::atomic<uintptr_t> g_live_lo[LIVE_PAIRS], g_live_hi[LIVE_PAIRS];
std::atomic<int> g_live_count{0};
void live_pair_add(const char* va, size_t bytes)
{
    const uintptr_t lo = reinterpret_cast<uintptr_t>(va);
    for (int i = 0; i < LIVE_PAIRS; ++i) {
        uintptr_t none = 0;
        if (g_live_hi[i].load(std::memory_order_relaxed) == 0 && g_live_lo[i].compare_exchange_strong(none, lo, std::memory_order_acq_rel)) {
            g_live_hi[i].store(lo + bytes, std::memory_order_release);
            g_live_count.fetch_add(1, std::memory_order_release);
            return;
        }
    }                                                           // more than 32 live pairs: the later ones launch by the table
}
void live_pair_remove(const char* va)
{
    const uintptr_t lo = reinterpret_cast<uintptr_t>(va);
    for (int i = 0; i < LIVE_PAIRS; ++i)
        if (g_live_lo[i].load(std::memory_order_acquire) == lo && g_live_hi[i].load(std::memory_order_acquire) != 0) {
            g_live_hi[i].store(0, std::memory_order_release);
            g_live_lo[i].store(0, std::memory_order_release);
            g_live_count.fetch_sub(1, std::memory_order_release);
            return;
        }
}

// at least two of a call's pointers inside one live constructed pair (its input and its output; widths[] / offsets[] / references may live anywhere)
bool fl_in_constructed_pair(std::initializer_list<const void*> ptrs)
{
    if (g_live_count.load(std::memory_order_acquire) == 0) return false;
    for (int i = 0; i < LIVE_PAIRS; ++i) {
        const uintptr_t hi = g_live_hi[i].load(std::memory_order_acquire), lo = g_live_lo[i].load(std::memory_order_acquire);
        if (!hi) continue;
        int inside = 0;
        for (const void* p : ptrs) {
            const uintptr_t a = reinterpret_cast<uintptr_t>(p);
            inside += (a >= lo && a < hi);
        }
        if (inside >= 2) return true;
    }
    return false;
}

// (one construction at a time per process: the class probe TIMES small kernels, and two threads probing at once -- on one device or on two
// that share nothing but this code -- would read each other's interference as class boundaries)
std::mutex g_construct_mutex;

hipError_t pair_alloc_interleaved(size_t in_bytes, size_t aux_bytes, size_t out_bytes, hipStream_t s, ColumnPair& p, int& rc)
{
    std::lock_guard<std::mutex> one_at_a_time(g_construct_mutex);
    rc = FL_OK;
    int dev = 0, vmm = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev);
    if (e != hipSuccess) return e;
    if (!vmm) return hipErrorNotSupported;
    const size_t in_span = pair_pad(in_bytes) + pair_pad(aux_bytes), total = in_span + pair_pad(out_bytes);
    const size_t chunk = PAIR_GIB;        // the class probe reads a whole piece: 0.17 ms per GiB is cleanly binary, 256-MiB pieces (45 us) are not
    const size_t n_in = std::max<size_t>(1, (in_span + chunk - 1) / chunk), n_out = std::max<size_t>(1, (pair_pad(out_bytes) + chunk - 1) / chunk);
    // enough chunks that a third of them holds the input and the other two thirds hold half the output each -- and twice the pair, because
    // the classes come in clusters of 4 .. 16 chunks (profiles/r06_vmm_placement.txt): a pool of just the pair's size often lacks one class
    // ... and never less than 48 GiB of them: a small pair's pool would otherwise lie inside one or two clusters
    // (with the output rotating through all three classes the input's class carries n_in + n_out / 3 of them)
    const int out_classes = pair_pad(out_bytes) >= 3 * in_span ? 2 : 3;
    size_t n_pool = std::max(std::max(std::max(3 * n_in + (out_classes == 3 ? n_out : 0), (3 * n_out + 1) / 2), 2 * (n_in + n_out)), 48 * PAIR_GIB / chunk);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 3 * PAIR_GIB) n_pool = std::min(n_pool, (free_b - 2 * PAIR_GIB) / chunk);
    if (n_pool < n_in + n_out) return hipErrorOutOfMemory;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> pool;
    auto drop_pool = [&] { for (auto h : pool) recycle_chunk(h, dev); pool.clear(); };
    p.dev = dev;
    const size_t pool_cap = free_b > 3 * PAIR_GIB ? (free_b - 2 * PAIR_GIB) / chunk : n_pool;
    std::vector<int> cls, order;
    // The pool GROWS while the arrangement it allows is poor: the classes come in clusters of 4 .. 32 chunks, so a pool of the first size
    // is sometimes two thirds one class (a box of round 6: in A x10 | out BAAAABBCAAABBAA -- Delta over a u8 column at 0.77 where a
    // balanced pool gives 0.80).  A round = more chunks (half as many again, three rounds at most, never beyond three times the pair or the free memory), ALL of
    // them classified again through a fresh scratch range (2.4 ms per chunk), the arrangement chosen again; "poor" = fewer than nine in
    // ten output positions got the class the rotation asks for.
    for (int round = 0;; ++round) {
        pool.reserve(n_pool);
        while (pool.size() < n_pool) {
            hipMemGenericAllocationHandle_t h;
            if (!cached_chunk(dev, h) && hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
            pool.push_back(h);
        }
        if (pool.size() < n_in + n_out) { drop_pool(); return hipErrorOutOfMemory; }
        // 1. every chunk's class, measured through a scratch address range (used once, never again)
        cls.assign(pool.size(), -1);
        char* scratch = nullptr;
        e = reserve_fresh_range(pool.size() * chunk, &scratch);
        if (e != hipSuccess) { drop_pool(); return e; }
        size_t mapped = 0;
        for (; mapped < pool.size() && e == hipSuccess; ++mapped) e = hipMemMap(scratch + mapped * chunk, chunk, 0, pool[mapped], 0);
        if (e != hipSuccess) --mapped;
        if (e == hipSuccess) e = hipMemSetAccess(scratch, pool.size() * chunk, &acc, 1);
        if (e == hipSuccess) rc = probe_classes(scratch, pool.size(), chunk, cls.data(), s);
        if (e == hipSuccess && rc == FL_OK) e = hipStreamSynchronize(s);
        for (size_t i = 0; i < mapped; ++i) (void)hipMemUnmap(scratch + i * chunk, chunk);
        (void)hipMemAddressFree(scratch, pool.size() * chunk);
        if (e != hipSuccess || rc != FL_OK) { drop_pool(); return e; }
        // 2. the pair's chunks in their final order
        size_t kept = 0;
        choose_chunks(cls, n_in, n_out, out_classes, order, &kept);
        const bool complete = order.size() == n_in + n_out;
        const size_t bigger = std::min(std::min(pool_cap, std::max<size_t>(3 * (n_in + n_out), 96)), pool.size() + std::max<size_t>(pool.size() / 2, 8));
        if ((complete && 10 * kept >= 9 * n_out) || round == 3 || pool.size() < n_pool || bigger <= pool.size()) {
            if (!complete) { drop_pool(); return hipErrorOutOfMemory; }
            break;
        }
        n_pool = bigger;
    }
    // ... mapped ONCE into the address range the caller gets
    std::vector<char> keep(pool.size(), 0);
    for (int g : order) keep[g] = 1;
    p.chunk_bytes = chunk;
    p.va_bytes = order.size() * chunk;
    e = reserve_fresh_range(p.va_bytes, &p.va);
    for (size_t i = 0; i < order.size() && e == hipSuccess; ++i) {
        e = hipMemMap(p.va + i * chunk, chunk, 0, pool[order[i]], 0);
        if (e == hipSuccess) p.n_mapped = i + 1;
    }
    if (e == hipSuccess) e = hipMemSetAccess(p.va, p.va_bytes, &acc, 1);
    for (size_t g = 0; g < pool.size(); ++g) {
        if (keep[g]) p.chunks.push_back(pool[g]);
        else recycle_chunk(pool[g], dev);
    }
    pool.clear();
    if (e != hipSuccess) { p.release(); return e; }
    for (size_t i = 0; i < order.size() && i + 1 < sizeof p.class_map; ++i) p.class_map[i] = cls[order[i]] < 0 ? '?' : (char)('A' + cls[order[i]]);
    p.in = p.va;
    p.aux = aux_bytes ? p.va + pair_pad(in_bytes) : nullptr;
    p.out = p.va + n_in * chunk;
    live_pair_add(p.va, p.va_bytes);
    return hipSuccess;
}

hipError_t pair_alloc(int layout, size_t in_bytes, size_t aux_bytes, size_t out_bytes, hipStream_t s, ColumnPair& p, int& rc)
{
    rc = FL_OK;
    if (layout == FL_LAYOUT_INTERLEAVED && pair_pad(in_bytes) + pair_pad(aux_bytes) + pair_pad(out_bytes) >= PAIR_INTERLEAVED_MIN)
        return pair_alloc_interleaved(in_bytes, aux_bytes, out_bytes, s, p, rc);
    if (layout == FL_LAYOUT_SEPARATE || layout == FL_LAYOUT_INTERLEAVED) {
        hipError_t e = hipMalloc(&p.bufs[0], in_bytes ? in_bytes : PAIR_ALIGN);
        if (e == hipSuccess && aux_bytes) e = hipMalloc(&p.bufs[1], aux_bytes);
        if (e == hipSuccess) e = hipMalloc(&p.bufs[2], out_bytes ? out_bytes : PAIR_ALIGN);
        if (e != hipSuccess) { p.release(); return e; }
        p.in = p.bufs[0]; p.aux = p.bufs[1]; p.out = p.bufs[2];
        return hipSuccess;
    }
    // ONE allocation: the input (then aux) at offset 0, the output centred on the first 64-GiB multiple that leaves room for them
    const size_t in_end = pair_pad(in_bytes) + pair_pad(aux_bytes), half = pair_pad(out_bytes) / 2;
    size_t k = 1;
    while (k * PAIR_ZONE < half + in_end) {
        if (++k > 8) return hipErrorOutOfMemory;
    }
    const size_t out_off = (k * PAIR_ZONE - half) & ~(PAIR_ALIGN - 1);
    hipError_t e = hipMalloc(&p.bufs[0], out_off + pair_pad(out_bytes));
    if (e != hipSuccess) return e;
    char* base = static_cast<char*>(p.bufs[0]);
    p.in = base;
    p.aux = aux_bytes ? base + pair_pad(in_bytes) : nullptr;
    p.out = base + out_off;
    return hipSuccess;
}

// GB/s of a bare stream in -> out over the pair (median of 5 after 2 untimed), in the proportion of the two sizes; 0 = not measured
int pair_probe(const ColumnPair& p, size_t in_bytes, size_t aux_bytes, size_t out_bytes, hipStream_t s, double& gbps)
{
    const size_t big = in_bytes > out_bytes ? in_bytes : out_bytes;
    const size_t n_units = big / 4096;
    gbps = 0.0;
    if (n_units < 1024) return FL_OK;                           // too small to say anything
    auto unit = [&](size_t bytes) { size_t u = bytes / n_units & ~(size_t)15; return u > 4096 ? 4096 : u; };
    BareArgs a{static_cast<const char*>(p.in), static_cast<const char*>(p.aux), static_cast<char*>(p.out), n_units, 0,
               (unsigned)unit(in_bytes), aux_bytes >= n_units * 128 ? 128u : 0u, (unsigned)unit(out_bytes), 63u};
    if (a.out_unit == 0) return FL_OK;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    hipError_t e = hipEventCreate(&t0);
    if (e == hipSuccess) e = hipEventCreate(&t1);
    float ms[5] = {0, 0, 0, 0, 0};
    for (int i = -2; i < 5 && e == hipSuccess; ++i) {
        e = hipEventRecord(t0, s);
        if (e == hipSuccess) e = launch_bare_stream(a, true, a.in_unit > a.out_unit ? 8 : 6, WINDOW_WHOLE, s);
        if (e == hipSuccess) e = hipEventRecord(t1, s);
        if (e == hipSuccess) e = hipEventSynchronize(t1);
        float x = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&x, t0, t1);
        if (i >= 0) ms[i] = x;
    }
    if (t0) (void)hipEventDestroy(t0);
    if (t1) (void)hipEventDestroy(t1);
    if (e != hipSuccess) return hip_fail(e);
    std::sort(ms, ms + 5);
    if (!(ms[2] > 0.f)) return FL_OK;                           // a zero median: not measured
    gbps = (double)n_units * (a.in_unit + a.aux_unit + a.out_unit) / (ms[2] * 1e6);
    return FL_OK;
}
}  // namespace

int fl_column_pair_alloc(size_t in_bytes, size_t aux_bytes, size_t out_bytes, int layout, void* stream, void** in, void** aux, void** out,
                         void** handle, int* layout_kept, uint32_t* probe_gbps)
{
    if (!in || !out || !handle || (aux_bytes && !aux)) return FL_ERR_NULL;
    if (layout < 0 || layout >= FL_LAYOUT_COUNT) return FL_ERR_INDEX;
    FL_DEVICE_TIER(stream);                                  // FL_CHECK_DEVICE=1: `stream` must belong to the current device (the memory will)
    hipStream_t s = static_cast<hipStream_t>(stream);
    *in = *out = *handle = nullptr;
    if (aux) *aux = nullptr;
    if (probe_gbps) for (int i = 0; i < FL_LAYOUT_COUNT; ++i) probe_gbps[i] = 0;
    ColumnPair* kept = new (std::nothrow) ColumnPair;
    if (!kept) return hip_fail(hipErrorOutOfMemory);
    int kept_layout = layout, rc = FL_OK;
    const size_t biggest = in_bytes > out_bytes ? in_bytes : out_bytes;
    if (layout == FL_LAYOUT_PROBE && biggest / 4096 < 1024) layout = kept_layout = FL_LAYOUT_SEPARATE;   // too small to time: nothing to choose
    if (layout != FL_LAYOUT_PROBE) {
        if (hipError_t e = pair_alloc(layout, in_bytes, aux_bytes, out_bytes, s, *kept, rc); e != hipSuccess || rc != FL_OK) {
            kept->release();
            delete kept;
            return rc != FL_OK ? rc : hip_fail(e);
        }
    } else {
        // the candidates one after the other (one that cannot be allocated next to the pair already held is skipped): a bare stream of
        // the pair's read : write proportion is timed on each; a later candidate replaces the kept one only if it wins by a margin
        // (ZONED pins ~64 GiB: 2 %).  The contents of the buffers are whatever the stream left there.
        double best = -1.0;
        kept_layout = -1;
        const bool large = pair_pad(in_bytes) + pair_pad(aux_bytes) + pair_pad(out_bytes) >= PAIR_INTERLEAVED_MIN;
        for (int cand : {FL_LAYOUT_INTERLEAVED, FL_LAYOUT_SEPARATE, FL_LAYOUT_ZONED}) {
            if (cand == FL_LAYOUT_INTERLEAVED && !large) continue;      // would be the SEPARATE candidate twice
            ColumnPair p;
            int prc = FL_OK;
            if (pair_alloc(cand, in_bytes, aux_bytes, out_bytes, s, p, prc) != hipSuccess || prc != FL_OK) { (void)hipGetLastError(); p.release(); continue; }
            double gbps = 0.0;
            if (const int r = pair_probe(p, in_bytes, aux_bytes, out_bytes, s, gbps); r != FL_OK) {
                p.release(); kept->release(); delete kept; return r;
            }
            if (probe_gbps) probe_gbps[cand] = (uint32_t)(gbps + 0.5);
            const double margin = cand == FL_LAYOUT_ZONED ? 1.02 : 1.01;
            if (kept_layout < 0 || gbps > best * margin) { kept->release(); *kept = std::move(p); p = ColumnPair(); best = gbps; kept_layout = cand; }
            else p.release();
        }
        if (kept_layout < 0) { delete kept; return hip_fail(hipErrorOutOfMemory); }
    }
    *in = kept->in;
    if (aux) *aux = kept->aux;
    *out = kept->out;
    *handle = kept;
    if (layout_kept) *layout_kept = kept_layout;
    return FL_OK;
}

int fl_column_pair_free(void* handle)
{
    if (!handle) return FL_OK;
    ColumnPair* p = static_cast<ColumnPair*>(handle);
    p->release();
    delete p;
    return FL_OK;
}

int fl_internal_selftune_check(int op, unsigned type_bits, unsigned width, const void* in, const void* aux, void* out, size_t n_blocks, void* stream,
                               float* table_ms, float* best_other_ms, int* best_other_policy)
{
    if (!table_ms || !best_other_ms || !best_other_policy) return FL_ERR_NULL;
    if (op < 0 || op > 2 || (type_bits != 8 && type_bits != 16 && type_bits != 32 && type_bits != 64)) return FL_ERR_INDEX;
    if (width > type_bits) return FL_ERR_WIDTH;
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto call = [&]() -> int {
        switch (type_bits * 4 + (unsigned)op) {
        case 8 * 4 + 0: return fl_u8_unpack(width, (const uint8_t*)in, (uint8_t*)out, n_blocks, stream);
        case 8 * 4 + 1: return fl_u8_pack(width, (const uint8_t*)in, (uint8_t*)out, n_blocks, stream);
        case 8 * 4 + 2: return fl_u8_undelta_pack(width, (const uint8_t*)in, (const uint8_t*)aux, (uint8_t*)out, n_blocks, stream);
        case 16 * 4 + 0: return fl_u16_unpack(width, (const uint16_t*)in, (uint16_t*)out, n_blocks, stream);
        case 16 * 4 + 1: return fl_u16_pack(width, (const uint16_t*)in, (uint16_t*)out, n_blocks, stream);
        case 16 * 4 + 2: return fl_u16_undelta_pack(width, (const uint16_t*)in, (const uint16_t*)aux, (uint16_t*)out, n_blocks, stream);
        case 32 * 4 + 0: return fl_u32_unpack(width, (const uint32_t*)in, (uint32_t*)out, n_blocks, stream);
        case 32 * 4 + 1: return fl_u32_pack(width, (const uint32_t*)in, (uint32_t*)out, n_blocks, stream);
        case 32 * 4 + 2: return fl_u32_undelta_pack(width, (const uint32_t*)in, (const uint32_t*)aux, (uint32_t*)out, n_blocks, stream);
        case 64 * 4 + 0: return fl_u64_unpack(width, (const uint64_t*)in, (uint64_t*)out, n_blocks, stream);
        case 64 * 4 + 1: return fl_u64_pack(width, (const uint64_t*)in, (uint64_t*)out, n_blocks, stream);
        default: return fl_u64_undelta_pack(width, (const uint64_t*)in, (const uint64_t*)aux, (uint64_t*)out, n_blocks, stream);
        }
    };
    hipEvent_t t0 = nullptr, t1 = nullptr;
    hipError_t e = hipEventCreate(&t0);
    if (e == hipSuccess) e = hipEventCreate(&t1);
    if (e != hipSuccess) { if (t0) (void)hipEventDestroy(t0); return hip_fail(e); }
    const int saved = fl_internal_get_kernel_policy();
    int rc = FL_OK;
    auto timed = [&](int policy, float& ms) {
        fl_internal_set_kernel_policy(policy);
        float t[3] = {0.f, 0.f, 0.f};
        for (int i = -1; i < 3 && rc == FL_OK; ++i) {
            hipError_t h = hipEventRecord(t0, s);
            if (h == hipSuccess) rc = call();
            if (h == hipSuccess && rc == FL_OK) h = hipEventRecord(t1, s);
            if (h == hipSuccess && rc == FL_OK) h = hipEventSynchronize(t1);
            float x = 0.f;
            if (h == hipSuccess && rc == FL_OK) h = hipEventElapsedTime(&x, t0, t1);
            if (h != hipSuccess) rc = hip_fail(h);
            if (i >= 0) t[i] = x;
        }
        std::sort(t, t + 3);
        ms = t[1];
    };
    timed(0, *table_ms);
    *best_other_ms = 0.f;
    *best_other_policy = 0;
    const fl::WaveOp wop = op == 1 ? fl::WAVE_PACK : op == 2 ? fl::WAVE_UNDELTA_PACK : fl::WAVE_UNPACK;
    int tab = fl::wave_policy(type_bits, width, wop);
    if (tab >= fl::TWO_BLOCKS) tab -= fl::TWO_BLOCKS;
    for (int policy : {1, 2 + 256 * 3, 2 + 256 * 4, 2 + 256 * 5, 2 + 256 * 6, 2 + 256 * 8}) {
        if (rc != FL_OK) break;
        if (policy == 1 && (tab == 0 || !fl::cell_column_built(type_bits, width, wop))) continue;   // the table's own choice, or not built
        if (policy != 1 && (policy >> 8) == tab && fl::wave_policy(type_bits, width, wop) < fl::TWO_BLOCKS) continue;   // the table's own choice
        float ms = 0.f;
        timed(policy, ms);
        if (rc == FL_OK && ms > 0.f && (*best_other_ms == 0.f || ms < *best_other_ms)) { *best_other_ms = ms; *best_other_policy = policy; }
    }
    fl_internal_set_kernel_policy(saved);
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    return rc;
}

size_t fl_internal_pair_chunk_cache(size_t max_chunks)
{
    std::vector<hipMemGenericAllocationHandle_t> drop;
    size_t held = 0;
    while (g_chunk_cache_lock.test_and_set(std::memory_order_acquire)) {}
    g_chunk_cache_limit = max_chunks;
    for (auto& c : g_chunk_cache) {
        while (c.size() > max_chunks) { drop.push_back(c.back()); c.pop_back(); }
        held += c.size();
    }
    g_chunk_cache_lock.clear(std::memory_order_release);
    for (auto h : drop) (void)hipMemRelease(h);
    return held;
}

size_t fl_internal_choose_chunks(const int* classes, size_t n_pool, size_t n_in, size_t n_out, int out_classes, int* order)
{
    if (!classes || !order || n_pool == 0) return 0;
    std::vector<int> cls(classes, classes + n_pool), chosen;
    choose_chunks(cls, n_in, n_out, out_classes, chosen);
    for (size_t i = 0; i < chosen.size(); ++i) order[i] = chosen[i];
    return chosen.size();
}

const char* fl_internal_column_pair_classes(const void* handle)
{
    return handle ? static_cast<const ColumnPair*>(handle)->class_map : "";
}

void fl_mixed_plan_destroy(fl_mixed_plan* p)
{
    if (!p) return;
    if (p->d_widths) (void)hipFree(p->d_widths);
    if (p->d_offsets) (void)hipFree(p->d_offsets);
    delete p;
}
size_t fl_mixed_plan_n_blocks(const fl_mixed_plan* p) { return p ? p->n_blocks : 0; }
uint64_t fl_mixed_plan_packed_bytes(const fl_mixed_plan* p) { return p ? p->packed_bytes : 0; }
const uint64_t* fl_mixed_plan_offsets(const fl_mixed_plan* p) { return p ? p->d_offsets : nullptr; }
const uint8_t* fl_mixed_plan_widths(const fl_mixed_plan* p) { return p ? p->d_widths : nullptr; }

void fl_host_release(void) { g_host.release(); }
void fl_internal_set_kernel_policy(int policy)
{
    const int mode = policy & 0xff, waves = (policy >> 8) & 0xff, bpw = (policy >> 16) & 0xff, prefetch = (policy >> 24) & 1,
              window = (policy >> 25) & 31;
    const bool ok = policy >= 0 && policy < (1 << 30) && mode <= 2 && (waves == 0 || (waves >= 3 && waves <= 8)) && bpw <= 16
                    && (mode == 2 || (waves == 0 && bpw == 0)) && (prefetch == 0 || bpw >= 2) && (window == 0 || window >= 8);
    g_kernel_policy.store(ok ? policy : 0, std::memory_order_relaxed);
    fl::window_override().store(ok ? window : 0, std::memory_order_relaxed);
}
int fl_internal_get_kernel_policy(void) { return g_kernel_policy.load(std::memory_order_relaxed); }
uint64_t fl_internal_zero_copy_fallbacks(void) { return g_zero_copy_fallbacks.load(std::memory_order_relaxed); }

#ifdef FL_ALL_CELL_COLUMN
const char* fl_version(void) { return "fastlanes_amd 0.6.0 (gfx950; wire format of spiraldb/fastlanes 0.1.8; FULL build: every cell-column instance, for A/B sweeps)"; }
#else
const char* fl_version(void) { return "fastlanes_amd 0.6.0 (gfx950; wire format of spiraldb/fastlanes 0.1.8)"; }
#endif

const char* fl_status_string(int status)
{
    switch (status) {
    case FL_OK: return "ok";
    case FL_ERR_WIDTH: return "width > T";
    case FL_ERR_INDEX: return "index out of range";
    case FL_ERR_NULL: return "null pointer";
    case FL_ERR_ALIGN: return "device pointer (or offset / size) not aligned as required";
    case FL_ERR_HIP: return "HIP runtime error";
    case FL_ERR_BOUNDS: return "block outside the packed column";
    case FL_ERR_DEVICE: return "pointer or stream does not belong to the current device (FL_CHECK_DEVICE)";
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
    int fl_##S##_pack(unsigned w, const T* in, T* out, size_t n, void* s) { FL_DEVICE_TIER(s, in, out); return dev_pack<T>(w, in, out, n, s); } \
    int fl_##S##_unpack(unsigned w, const T* in, T* out, size_t n, void* s) { FL_DEVICE_TIER(s, in, out); return dev_unpack<T>(w, in, out, n, s); } \
    int fl_##S##_unpack_single(unsigned w, const T* pk, size_t n, const uint64_t* idx, size_t ni, T* out,  \
                               uint32_t* ef, void* s)                                                     \
    { FL_DEVICE_TIER(s, pk, idx, out, ef); return dev_unpack_single<T>(w, pk, n, idx, ni, out, ef, s); }                                       \
    int fl_##S##_for_pack(unsigned w, const T* in, const T* r, size_t rs, T* out, size_t n, void* s)      \
    { FL_DEVICE_TIER(s, in, r, out); return dev_for_pack<T>(w, in, r, rs, out, n, s); }                                                  \
    int fl_##S##_unfor_pack(unsigned w, const T* in, const T* r, size_t rs, T* out, size_t n, void* s)    \
    { FL_DEVICE_TIER(s, in, r, out); return dev_unfor_pack<T>(w, in, r, rs, out, n, s); }                                                \
    int fl_##S##_delta(const T* in, const T* b, T* out, size_t n, void* s) { FL_DEVICE_TIER(s, in, b, out); return dev_delta<T>(false, in, b, out, n, s); } \
    int fl_##S##_undelta(const T* in, const T* b, T* out, size_t n, void* s) { FL_DEVICE_TIER(s, in, b, out); return dev_delta<T>(true, in, b, out, n, s); } \
    int fl_##S##_undelta_pack(unsigned w, const T* in, const T* b, T* out, size_t n, void* s)             \
    { FL_DEVICE_TIER(s, in, b, out); return dev_undelta_pack<T>(w, in, b, out, n, s); }                                                  \
    int fl_##S##_undelta_pack_untranspose(unsigned w, const T* in, const T* b, T* out, size_t n, void* s) \
    { FL_DEVICE_TIER(s, in, b, out); return dev_undelta_pack_untranspose<T>(w, in, b, out, n, s); }                                      \
    int fl_##S##_transpose_delta_pack(unsigned w, const T* in, const T* b, T* out, size_t n, void* s)     \
    { FL_DEVICE_TIER(s, in, b, out); return dev_transpose_delta_pack<T>(w, in, b, out, n, s); }                                          \
    int fl_##S##_unpack_block_sums(unsigned w, const T* in, size_t n, uint64_t* sums, void* s)           \
    { FL_DEVICE_TIER(s, in, sums); return dev_unpack_block_sums<T>(w, in, n, sums, s); }                                               \
    int fl_##S##_unpack_compare(unsigned w, const T* in, int op, T k, size_t n, uint32_t* mask, void* s)  \
    { FL_DEVICE_TIER(s, in, mask); return dev_unpack_compare<T>(w, in, op, k, n, mask, s); }                                           \
    int fl_##S##_block_min_max(const T* in, size_t n, T* mins, T* maxs, void* s)                          \
    { FL_DEVICE_TIER(s, in, mins, maxs); return dev_block_min_max<T>(in, n, mins, maxs, s); }                                                \
    int fl_##S##_transpose(const T* in, T* out, size_t n, void* s) { FL_DEVICE_TIER(s, in, out); return dev_transpose<T>(false, in, out, n, s); } \
    int fl_##S##_untranspose(const T* in, T* out, size_t n, void* s) { FL_DEVICE_TIER(s, in, out); return dev_transpose<T>(true, in, out, n, s); } \
    int fl_##S##_unpack_mixed(const fl_mixed_plan* p, const T* pk, T* out, void* s) { FL_DEVICE_TIER(s, pk, out); return run_mixed<T>(false, p, pk, out, s); } \
    int fl_##S##_pack_mixed(const fl_mixed_plan* p, const T* in, T* pk, void* s) { FL_DEVICE_TIER(s, in, pk); return run_mixed<T>(true, p, pk, const_cast<T*>(in), s); } \
    int fl_##S##_unpack_widths(const uint8_t* w, const uint64_t* o, const T* pk, size_t pb, T* out, size_t n, uint32_t* ef, void* s) \
    { FL_DEVICE_TIER(s, w, o, pk, out, ef); return run_widths<T>(false, w, o, pk, pb, out, n, ef, s); }                                         \
    int fl_##S##_pack_widths(const uint8_t* w, const uint64_t* o, const T* in, T* pk, size_t pb, size_t n, uint32_t* ef, void* s) \
    { FL_DEVICE_TIER(s, w, o, in, pk, ef); return run_widths<T>(true, w, o, pk, pb, const_cast<T*>(in), n, ef, s); }                           \
    int fl_##S##_unfor_pack_widths(const uint8_t* w, const uint64_t* o, const T* pk, size_t pb, const T* r, size_t rs, T* out, \
                                   size_t n, uint32_t* ef, void* s)                                        \
    { FL_DEVICE_TIER(s, w, o, pk, r, out, ef); return run_widths<T>(false, w, o, pk, pb, out, n, ef, s, r, rs, true); }                        \
    int fl_##S##_for_pack_widths(const uint8_t* w, const uint64_t* o, const T* in, const T* r, size_t rs, T* pk, size_t pb, \
                                 size_t n, uint32_t* ef, void* s)                                          \
    { FL_DEVICE_TIER(s, w, o, in, r, pk, ef); return run_widths<T>(true, w, o, pk, pb, const_cast<T*>(in), n, ef, s, r, rs, true); }          \
    int fl_##S##_for_widths(const T* mins, const T* maxs, size_t n, uint8_t* w, void* s)                   \
    { FL_DEVICE_TIER(s, mins, maxs, w); return dev_for_widths<T>(mins, maxs, n, w, s); }                                                      \
    int fl_##S##_undelta_pack_widths(const uint8_t* w, const uint64_t* o, const T* pk, size_t pb, const T* b, T* out, size_t n, \
                                     uint32_t* ef, void* s)                                                \
    { FL_DEVICE_TIER(s, w, o, pk, b, out, ef); return run_chain_widths<T>(OP_UNDELTA_PACK, w, o, pk, b, out, pb, n, ef, s); }                  \
    int fl_##S##_undelta_pack_untranspose_widths(const uint8_t* w, const uint64_t* o, const T* pk, size_t pb, const T* b, T* out, \
                                                 size_t n, uint32_t* ef, void* s)                          \
    { FL_DEVICE_TIER(s, w, o, pk, b, out, ef); return run_chain_widths<T>(OP_UNDELTA_PACK_UNTRANSPOSE, w, o, pk, b, out, pb, n, ef, s); }      \
    int fl_##S##_transpose_delta_pack_widths(const uint8_t* w, const uint64_t* o, const T* in, const T* b, T* pk, size_t pb, \
                                             size_t n, uint32_t* ef, void* s)                              \
    { FL_DEVICE_TIER(s, w, o, in, b, pk, ef); return run_chain_widths<T>(OP_TRANSPOSE_DELTA_PACK, w, o, in, b, pk, pb, n, ef, s); }            \
    int fl_##S##_unpack_batch(const T* const* pk, T* const* out, const uint8_t* w, const uint32_t* nb, size_t na, uint32_t mb, \
                              uint32_t* ef, void* s)                                                      \
    { FL_DEVICE_TIER(s, pk, out, w, nb, ef); return run_batch<T>(false, reinterpret_cast<const void* const*>(pk), reinterpret_cast<void* const*>(out), w, nullptr, false, nb, na, mb, ef, s); } \
    int fl_##S##_unfor_pack_batch(const T* const* pk, T* const* out, const uint8_t* w, const T* refs, const uint32_t* nb, size_t na, \
                                  uint32_t mb, uint32_t* ef, void* s)                                     \
    { FL_DEVICE_TIER(s, pk, out, w, refs, nb, ef); return run_batch<T>(false, reinterpret_cast<const void* const*>(pk), reinterpret_cast<void* const*>(out), w, refs, true, nb, na, mb, ef, s); } \
    int fl_##S##_for_pack_batch(const T* const* in, T* const* pk, const uint8_t* w, const T* refs, const uint32_t* nb, size_t na, \
                                uint32_t mb, uint32_t* ef, void* s)                                       \
    { FL_DEVICE_TIER(s, pk, in, w, refs, nb, ef); return run_batch<T>(true, reinterpret_cast<const void* const*>(pk), (void* const*)in, w, refs, true, nb, na, mb, ef, s); } \
    int fl_##S##_pack_batch(const T* const* in, T* const* pk, const uint8_t* w, const uint32_t* nb, size_t na, uint32_t mb, \
                            uint32_t* ef, void* s)                                                        \
    { FL_DEVICE_TIER(s, pk, in, w, nb, ef); return run_batch<T>(true, reinterpret_cast<const void* const*>(pk), (void* const*)in, w, nullptr, false, nb, na, mb, ef, s); } \
    int fl_##S##_undelta_pack_batch(const T* const* pk, const T* const* bs, T* const* out, const uint8_t* w, const uint32_t* nb, \
                                    size_t na, uint32_t mb, int untranspose, uint32_t* ef, void* s)        \
    { FL_DEVICE_TIER(s, pk, bs, out, w, nb, ef); return run_batch_chain<T>(untranspose ? OP_UNDELTA_PACK_UNTRANSPOSE : OP_UNDELTA_PACK, reinterpret_cast<const void* const*>(pk), reinterpret_cast<const void* const*>(bs), reinterpret_cast<void* const*>(out), w, nb, na, mb, ef, s); } \
    int fl_##S##_transpose_delta_pack_batch(const T* const* in, const T* const* bs, T* const* pk, const uint8_t* w, const uint32_t* nb, \
                                            size_t na, uint32_t mb, uint32_t* ef, void* s)                 \
    { FL_DEVICE_TIER(s, in, bs, pk, w, nb, ef); return run_batch_chain<T>(OP_TRANSPOSE_DELTA_PACK, reinterpret_cast<const void* const*>(pk), reinterpret_cast<const void* const*>(bs), (void* const*)in, w, nb, na, mb, ef, s); } \
    int fl_##S##_unpack_single_widths(const uint8_t* w, const uint64_t* o, const T* pk, size_t pb, size_t n, const uint64_t* idx, \
                                      size_t ni, T* out, uint32_t* ef, void* s)                           \
    { FL_DEVICE_TIER(s, w, o, pk, idx, out, ef); return dev_unpack_single_widths<T>(w, o, pk, pb, n, idx, ni, out, ef, s); }                         \
    int fl_##S##_pack_host(unsigned w, const T* in, T* out, size_t n)                                     \
    {                                                                                                     \
        if (w > sizeof(T) * 8) return FL_ERR_WIDTH;                                                       \
        return host_run<T>(in, n * 1024, nullptr, 0, out, n * plen<T>(w),                                 \
                           [&](const T* di, const T*, T* d_o, void* st) { return dev_pack<T>(w, di, d_o, n, st); }); \
    }                                                                                                     \
    int fl_##S##_unpack_host(unsigned w, const T* in, T* out, size_t n)                                   \
    {                                                                                                     \
        if (w > sizeof(T) * 8) return FL_ERR_WIDTH;                                                       \
        return host_run<T>(in, n * plen<T>(w), nullptr, 0, out, n * 1024,                                 \
                           [&](const T* di, const T*, T* d_o, void* st) { return dev_unpack<T>(w, di, d_o, n, st); }); \
    }                                                                                                     \
    int fl_##S##_unpack_single_host(unsigned w, const T* pk, size_t n, uint64_t index, T* value)          \
    { return host_unpack_single<T>(w, pk, n, index, value); }                                             \
    int fl_##S##_for_pack_host(unsigned w, const T* in, T reference, T* out, size_t n)                    \
    {                                                                                                     \
        if (w > sizeof(T) * 8) return FL_ERR_WIDTH;                                                       \
        return host_run<T>(in, n * 1024, &reference, 1, out, n * plen<T>(w),                              \
                           [&](const T* di, const T* da, T* d_o, void* st) { return dev_for_pack<T>(w, di, da, 0, d_o, n, st); }); \
    }                                                                                                     \
    int fl_##S##_unfor_pack_host(unsigned w, const T* in, T reference, T* out, size_t n)                  \
    {                                                                                                     \
        if (w > sizeof(T) * 8) return FL_ERR_WIDTH;                                                       \
        return host_run<T>(in, n * plen<T>(w), &reference, 1, out, n * 1024,                              \
                           [&](const T* di, const T* da, T* d_o, void* st) { return dev_unfor_pack<T>(w, di, da, 0, d_o, n, st); }); \
    }                                                                                                     \
    int fl_##S##_delta_host(const T* in, const T* b, T* out, size_t n)                                    \
    {                                                                                                     \
        return host_run<T>(in, n * 1024, b, n * (1024 / (sizeof(T) * 8)), out, n * 1024,                  \
                           [&](const T* di, const T* da, T* d_o, void* st) { return dev_delta<T>(false, di, da, d_o, n, st); }); \
    }                                                                                                     \
    int fl_##S##_undelta_host(const T* in, const T* b, T* out, size_t n)                                  \
    {                                                                                                     \
        return host_run<T>(in, n * 1024, b, n * (1024 / (sizeof(T) * 8)), out, n * 1024,                  \
                           [&](const T* di, const T* da, T* d_o, void* st) { return dev_delta<T>(true, di, da, d_o, n, st); }); \
    }                                                                                                     \
    int fl_##S##_undelta_pack_host(unsigned w, const T* in, const T* b, T* out, size_t n)                 \
    {                                                                                                     \
        if (w > sizeof(T) * 8) return FL_ERR_WIDTH;                                                       \
        return host_run<T>(in, n * plen<T>(w), b, n * (1024 / (sizeof(T) * 8)), out, n * 1024,            \
                           [&](const T* di, const T* da, T* d_o, void* st) { return dev_undelta_pack<T>(w, di, da, d_o, n, st); }); \
    }                                                                                                     \
    int fl_##S##_transpose_host(const T* in, T* out, size_t n)                                            \
    {                                                                                                     \
        return host_run<T>(in, n * 1024, nullptr, 0, out, n * 1024,                                       \
                           [&](const T* di, const T*, T* d_o, void* st) { return dev_transpose<T>(false, di, d_o, n, st); }); \
    }                                                                                                     \
    int fl_##S##_untranspose_host(const T* in, T* out, size_t n)                                          \
    {                                                                                                     \
        return host_run<T>(in, n * 1024, nullptr, 0, out, n * 1024,                                       \
                           [&](const T* di, const T*, T* d_o, void* st) { return dev_transpose<T>(true, di, d_o, n, st); }); \
    }

FL_DEFINE_TYPE(uint8_t, u8)
FL_DEFINE_TYPE(uint16_t, u16)
FL_DEFINE_TYPE(uint32_t, u32)
FL_DEFINE_TYPE(uint64_t, u64)

}  // extern "C"
