// fastlanes_amd.hpp -- C++ host-side mirror of the reference's codec traits over the
// C ABI (include/fastlanes_amd.h).  The reference is compiled Rust; with no Rust
// toolchain in the build image this header plays the role of the `impl BitPacking
// for u32 { .. }` shim (INTEGRATION.md shows the Rust form): same method names,
// argument meaning and error behaviour.
//
//   reference (Rust)                                    here (C++)
//   <T as BitPacking>::pack::<W>(&in, &mut out)         fastlanes::BitPacking<T>::pack<W>(in, out)
//   <T as BitPacking>::unchecked_pack(w, in, out)       fastlanes::BitPacking<T>::unchecked_pack(w, in, n_in, out, n_out)
//   <T as FoR>::for_pack::<W>(&in, r, &mut out)         fastlanes::FoR<T>::for_pack<W>(in, r, out)
//   <T as Delta>::undelta_pack::<W>(&in, &base, &mut o) fastlanes::Delta<T>::undelta_pack<W>(in, base, out)
//   <T as Transpose>::transpose(&in, &mut out)          fastlanes::Transpose<T>::transpose(in, out)
//
// Fixed-size array references enforce the sizes the Rust types enforce
// (bitpacking.rs:19,33); `W <= T` is a static_assert (bitpacking.rs:8-13).
// The `unchecked_` forms take slices (pointer + length) and, like the reference's
// debug_asserts (bitpacking.rs:78-80), check lengths -- here always.  Where the
// reference panics (width > T: bitpacking.rs:93; index >= 1024: :152) this throws
// fastlanes::Error.  Every call runs the gfx950 kernels (host slices are staged
// through HBM); there is no CPU implementation.  `*_device` members are the batched,
// device-resident form of the same methods (n_blocks contiguous blocks, async).
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <type_traits>

#include "fastlanes_amd.h"

namespace fastlanes {

constexpr std::size_t FL_ORDER[8] = {0, 4, 2, 6, 1, 5, 3, 7};   // lib.rs:22

struct Error : std::runtime_error {
    int status;
    Error(int st, const char* where)
        : std::runtime_error(std::string(where) + ": " + fl_status_string(st)), status(st) {}
};

namespace detail {
inline void check(int st, const char* where) { if (st != FL_OK) throw Error(st, where); }

template <typename T> struct Abi;
#define FASTLANES_ABI(T, S)                                                                          \
    template <> struct Abi<T> {                                                                      \
        static constexpr auto pack = fl_##S##_pack;                                                  \
        static constexpr auto unpack = fl_##S##_unpack;                                              \
        static constexpr auto unpack_single = fl_##S##_unpack_single;                                \
        static constexpr auto for_pack = fl_##S##_for_pack;                                          \
        static constexpr auto unfor_pack = fl_##S##_unfor_pack;                                      \
        static constexpr auto delta = fl_##S##_delta;                                                \
        static constexpr auto undelta = fl_##S##_undelta;                                            \
        static constexpr auto undelta_pack = fl_##S##_undelta_pack;                                  \
        static constexpr auto transpose = fl_##S##_transpose;                                        \
        static constexpr auto untranspose = fl_##S##_untranspose;                                    \
        static constexpr auto undelta_pack_untranspose = fl_##S##_undelta_pack_untranspose;          \
        static constexpr auto transpose_delta_pack = fl_##S##_transpose_delta_pack;                  \
        static constexpr auto unpack_block_sums = fl_##S##_unpack_block_sums;                        \
        static constexpr auto block_min_max = fl_##S##_block_min_max;                                \
        static constexpr auto unpack_compare = fl_##S##_unpack_compare;                              \
        static constexpr auto unpack_mixed = fl_##S##_unpack_mixed;                                  \
        static constexpr auto pack_mixed = fl_##S##_pack_mixed;                                      \
        static constexpr auto unpack_widths = fl_##S##_unpack_widths;                                \
        static constexpr auto pack_widths = fl_##S##_pack_widths;                                    \
        static constexpr auto unpack_single_widths = fl_##S##_unpack_single_widths;                  \
        static constexpr auto unfor_pack_widths = fl_##S##_unfor_pack_widths;                        \
        static constexpr auto for_pack_widths = fl_##S##_for_pack_widths;                            \
        static constexpr auto undelta_pack_widths = fl_##S##_undelta_pack_widths;                    \
        static constexpr auto undelta_pack_untranspose_widths = fl_##S##_undelta_pack_untranspose_widths; \
        static constexpr auto transpose_delta_pack_widths = fl_##S##_transpose_delta_pack_widths;    \
        static constexpr auto for_widths = fl_##S##_for_widths;                                      \
        static constexpr auto unpack_batch = fl_##S##_unpack_batch;                                  \
        static constexpr auto pack_batch = fl_##S##_pack_batch;                                      \
        static constexpr auto unfor_pack_batch = fl_##S##_unfor_pack_batch;                          \
        static constexpr auto for_pack_batch = fl_##S##_for_pack_batch;                              \
        static constexpr auto undelta_pack_batch = fl_##S##_undelta_pack_batch;                      \
        static constexpr auto transpose_delta_pack_batch = fl_##S##_transpose_delta_pack_batch;      \
        static constexpr auto pack_host = fl_##S##_pack_host;                                        \
        static constexpr auto unpack_host = fl_##S##_unpack_host;                                    \
        static constexpr auto unpack_single_host = fl_##S##_unpack_single_host;                      \
        static constexpr auto for_pack_host = fl_##S##_for_pack_host;                                \
        static constexpr auto unfor_pack_host = fl_##S##_unfor_pack_host;                            \
        static constexpr auto delta_host = fl_##S##_delta_host;                                      \
        static constexpr auto undelta_host = fl_##S##_undelta_host;                                  \
        static constexpr auto undelta_pack_host = fl_##S##_undelta_pack_host;                        \
        static constexpr auto transpose_host = fl_##S##_transpose_host;                              \
        static constexpr auto untranspose_host = fl_##S##_untranspose_host;                          \
    };
FASTLANES_ABI(std::uint8_t, u8)
FASTLANES_ABI(std::uint16_t, u16)
FASTLANES_ABI(std::uint32_t, u32)
FASTLANES_ABI(std::uint64_t, u64)
#undef FASTLANES_ABI
}  // namespace detail

// lib.rs:24-32
template <typename T> struct FastLanes {
    static constexpr std::size_t T_BITS = sizeof(T) * 8;   // `const T`
    static constexpr std::size_t LANES = 1024 / T_BITS;
};

// bitpacking.rs:16-59
template <typename T> struct BitPacking : FastLanes<T> {
    using A = detail::Abi<T>;
    static constexpr std::size_t TB = sizeof(T) * 8;

    template <std::size_t W> static void pack(const T (&input)[1024], T (&output)[W ? 1024 * W / TB : 1])
    {
        static_assert(W <= TB, "BitPackWidth<W>: SupportedBitPackWidth<T> (bitpacking.rs:8-13)");
        detail::check(A::pack_host(W, input, output, 1), "pack");
    }
    template <std::size_t W> static void unpack(const T (&input)[W ? 1024 * W / TB : 1], T (&output)[1024])
    {
        static_assert(W <= TB, "BitPackWidth<W>: SupportedBitPackWidth<T>");
        detail::check(A::unpack_host(W, input, output, 1), "unpack");
    }
    template <std::size_t W> static T unpack_single(const T (&packed)[W ? 1024 * W / TB : 1], std::size_t index)
    {
        static_assert(W <= TB, "BitPackWidth<W>: SupportedBitPackWidth<T>");
        T v{};
        detail::check(A::unpack_single_host(W, packed, 1, index, &v), "unpack_single");
        return v;
    }
    // bitpacking.rs:76-96 -- lengths as in the debug_asserts (:78-80)
    static void unchecked_pack(std::size_t width, const T* input, std::size_t in_len, T* output, std::size_t out_len)
    {
        if (width > TB) throw Error(FL_ERR_WIDTH, "unchecked_pack");
        if (in_len != 1024 || out_len != 128 * width / sizeof(T)) throw std::length_error("unchecked_pack: buffer sizes");
        detail::check(A::pack_host((unsigned)width, input, output, 1), "unchecked_pack");
    }
    // bitpacking.rs:109-129
    static void unchecked_unpack(std::size_t width, const T* input, std::size_t in_len, T* output, std::size_t out_len)
    {
        if (width > TB) throw Error(FL_ERR_WIDTH, "unchecked_unpack");
        if (out_len != 1024 || in_len != 128 * width / sizeof(T)) throw std::length_error("unchecked_unpack: buffer sizes");
        detail::check(A::unpack_host((unsigned)width, input, output, 1), "unchecked_unpack");
    }
    // bitpacking.rs:181-200
    static T unchecked_unpack_single(std::size_t width, const T* packed, std::size_t len, std::size_t index)
    {
        if (width > TB) throw Error(FL_ERR_WIDTH, "unchecked_unpack_single");
        if (len != 128 * width / sizeof(T)) throw std::length_error("unchecked_unpack_single: buffer size");
        T v{};
        detail::check(A::unpack_single_host((unsigned)width, packed, 1, index, &v), "unchecked_unpack_single");
        return v;
    }
    // batched, device-resident forms (stream = hipStream_t)
    static void pack_device(std::size_t width, const T* d_in, T* d_out, std::size_t n_blocks, void* stream = nullptr)
    { detail::check(A::pack((unsigned)width, d_in, d_out, n_blocks, stream), "pack_device"); }
    static void unpack_device(std::size_t width, const T* d_in, T* d_out, std::size_t n_blocks, void* stream = nullptr)
    { detail::check(A::unpack((unsigned)width, d_in, d_out, n_blocks, stream), "unpack_device"); }
    // extensions: reductions over what unpack / the unpacked block hold
    static void unpack_block_sums_device(std::size_t width, const T* d_packed, std::size_t n_blocks, std::uint64_t* d_sums,
                                         void* stream = nullptr)
    { detail::check(A::unpack_block_sums((unsigned)width, d_packed, n_blocks, d_sums, stream), "unpack_block_sums_device"); }
    // mask bit i of block b = (unpack(block b)[i] <op> constant); 32 words per block
    static void unpack_compare_device(std::size_t width, const T* d_packed, fl_cmp op, T constant, std::size_t n_blocks,
                                      std::uint32_t* d_mask, void* stream = nullptr)
    { detail::check(A::unpack_compare((unsigned)width, d_packed, (int)op, constant, n_blocks, d_mask, stream), "unpack_compare_device"); }
    static void block_min_max_device(const T* d_values, std::size_t n_blocks, T* d_mins, T* d_maxs, void* stream = nullptr)
    { detail::check(A::block_min_max(d_values, n_blocks, d_mins, d_maxs, stream), "block_min_max_device"); }
    static void unpack_single_device(std::size_t width, const T* d_packed, std::size_t n_blocks, const std::uint64_t* d_indices,
                                     std::size_t n_indices, T* d_out, std::uint32_t* d_err, void* stream = nullptr)
    { detail::check(A::unpack_single((unsigned)width, d_packed, n_blocks, d_indices, n_indices, d_out, d_err, stream), "unpack_single_device"); }
};

// ffor.rs:4-18
template <typename T> struct FoR : BitPacking<T> {
    using A = detail::Abi<T>;
    static constexpr std::size_t TB = sizeof(T) * 8;
    template <std::size_t W> static void for_pack(const T (&input)[1024], T reference, T (&output)[W ? 1024 * W / TB : 1])
    {
        static_assert(W <= TB, "BitPackWidth<W>: SupportedBitPackWidth<T>");
        detail::check(A::for_pack_host(W, input, reference, output, 1), "for_pack");
    }
    template <std::size_t W> static void unfor_pack(const T (&input)[W ? 1024 * W / TB : 1], T reference, T (&output)[1024])
    {
        static_assert(W <= TB, "BitPackWidth<W>: SupportedBitPackWidth<T>");
        detail::check(A::unfor_pack_host(W, input, reference, output, 1), "unfor_pack");
    }
    static void for_pack_device(std::size_t width, const T* d_in, const T* d_refs, std::size_t ref_stride, T* d_out,
                                std::size_t n_blocks, void* stream = nullptr)
    { detail::check(A::for_pack((unsigned)width, d_in, d_refs, ref_stride, d_out, n_blocks, stream), "for_pack_device"); }
    static void unfor_pack_device(std::size_t width, const T* d_in, const T* d_refs, std::size_t ref_stride, T* d_out,
                                  std::size_t n_blocks, void* stream = nullptr)
    { detail::check(A::unfor_pack((unsigned)width, d_in, d_refs, ref_stride, d_out, n_blocks, stream), "unfor_pack_device"); }
};

// delta.rs:6-17
template <typename T> struct Delta : BitPacking<T> {
    using A = detail::Abi<T>;
    static constexpr std::size_t TB = sizeof(T) * 8;
    static constexpr std::size_t LANES = 1024 / TB;
    static void delta(const T (&input)[1024], const T (&base)[LANES], T (&output)[1024])
    { detail::check(A::delta_host(input, base, output, 1), "delta"); }
    static void undelta(const T (&input)[1024], const T (&base)[LANES], T (&output)[1024])
    { detail::check(A::undelta_host(input, base, output, 1), "undelta"); }
    template <std::size_t W>
    static void undelta_pack(const T (&input)[W ? 1024 * W / TB : 1], const T (&base)[LANES], T (&output)[1024])
    {
        static_assert(W <= TB, "BitPackWidth<W>: SupportedBitPackWidth<T>");
        detail::check(A::undelta_pack_host(W, input, base, output, 1), "undelta_pack");
    }
    static void delta_device(const T* d_in, const T* d_bases, T* d_out, std::size_t n_blocks, void* stream = nullptr)
    { detail::check(A::delta(d_in, d_bases, d_out, n_blocks, stream), "delta_device"); }
    static void undelta_device(const T* d_in, const T* d_bases, T* d_out, std::size_t n_blocks, void* stream = nullptr)
    { detail::check(A::undelta(d_in, d_bases, d_out, n_blocks, stream), "undelta_device"); }
    static void undelta_pack_device(std::size_t width, const T* d_in, const T* d_bases, T* d_out, std::size_t n_blocks,
                                    void* stream = nullptr)
    { detail::check(A::undelta_pack((unsigned)width, d_in, d_bases, d_out, n_blocks, stream), "undelta_pack_device"); }
    // extensions: == untranspose(undelta_pack(..)) / pack(delta(transpose(..)))  (delta.rs:88-100)
    static void undelta_pack_untranspose_device(std::size_t width, const T* d_in, const T* d_bases, T* d_out,
                                                std::size_t n_blocks, void* stream = nullptr)
    { detail::check(A::undelta_pack_untranspose((unsigned)width, d_in, d_bases, d_out, n_blocks, stream), "undelta_pack_untranspose_device"); }
    static void transpose_delta_pack_device(std::size_t width, const T* d_in, const T* d_bases, T* d_out,
                                            std::size_t n_blocks, void* stream = nullptr)
    { detail::check(A::transpose_delta_pack((unsigned)width, d_in, d_bases, d_out, n_blocks, stream), "transpose_delta_pack_device"); }
};

// transpose.rs:4-7,29-36
constexpr std::size_t transpose(std::size_t idx)
{
    return (idx % 16) * 64 + FL_ORDER[(idx / 16) % 8] * 8 + idx / 128;
}
template <typename T> struct Transpose : FastLanes<T> {
    using A = detail::Abi<T>;
    static void transpose(const T (&input)[1024], T (&output)[1024])
    { detail::check(A::transpose_host(input, output, 1), "transpose"); }
    static void untranspose(const T (&input)[1024], T (&output)[1024])
    { detail::check(A::untranspose_host(input, output, 1), "untranspose"); }
    static void transpose_device(const T* d_in, T* d_out, std::size_t n_blocks, void* stream = nullptr)
    { detail::check(A::transpose(d_in, d_out, n_blocks, stream), "transpose_device"); }
    static void untranspose_device(const T* d_in, T* d_out, std::size_t n_blocks, void* stream = nullptr)
    { detail::check(A::untranspose(d_in, d_out, n_blocks, stream), "untranspose_device"); }
};

// ---------------------------------------------------------------------------------------------------------------------
// The reference's CALLER LOOP as one checked call.  Vortex (and benches/bitpacking.rs:80-97) hold a column as slices and loop
//     for i in 0..n { T::unchecked_unpack(w, &packed[i*pl..(i+1)*pl], &mut out[i*1024..(i+1)*1024]) }
// where every iteration asserts its two lengths (bitpacking.rs:78-80, :111-113).  A DeviceSlice is the device-resident
// counterpart of such a slice -- pointer AND element count -- and the *_column functions below check the same lengths for the
// whole column before the one asynchronous launch, so the fast path keeps what the slow path asserted.  A length mismatch
// throws std::length_error (the reference's debug_assert), width > T throws Error(FL_ERR_WIDTH) (its unreachable!()).
// (bindings/rust/src/device.rs is the same surface in Rust.)
// ---------------------------------------------------------------------------------------------------------------------
template <typename T> struct DeviceSlice {
    T* ptr = nullptr;          // device pointer, 16-byte aligned
    std::size_t len = 0;       // elements
    DeviceSlice() = default;
    DeviceSlice(T* p, std::size_t n) : ptr(p), len(n) {}
    template <typename U, typename = std::enable_if_t<std::is_same<const U, T>::value>>
    DeviceSlice(const DeviceSlice<U>& o) : ptr(o.ptr), len(o.len) {}          // DeviceSlice<T> -> DeviceSlice<const T>
    // elements [first, first + count): the sub-slice a sharded caller hands each device / stream
    DeviceSlice subslice(std::size_t first, std::size_t count) const
    {
        if (first > len || count > len - first) throw std::out_of_range("DeviceSlice::subslice");
        return DeviceSlice(ptr + first, count);
    }
};

namespace detail {
// number of 1024-value blocks of a column given as (packed, unpacked) slices of width `width`; throws unless
// unpacked.len == 1024 * n and packed.len == n * 1024 * width / T for one n  (bitpacking.rs:78-80, :111-113 for every block)
template <typename T>
inline std::size_t column_blocks(std::size_t width, std::size_t packed_len, std::size_t unpacked_len, const char* where)
{
    constexpr std::size_t TB = sizeof(T) * 8;
    if (width > TB) throw Error(FL_ERR_WIDTH, where);
    if (unpacked_len % 1024 != 0) throw std::length_error(std::string(where) + ": the unpacked slice must hold 1024 elements per block");
    const std::size_t n = unpacked_len / 1024;
    if (packed_len != n * (1024 * width / TB))
        throw std::length_error(std::string(where) + ": the packed slice must hold 1024 * W / T elements per block");
    return n;
}
}  // namespace detail

// for b in blocks { T::unchecked_unpack(width, &packed[b*pl..], &mut out[b*1024..]) }   (bitpacking.rs:109-129)
template <typename T>
inline void unpack_column(std::size_t width, DeviceSlice<const T> packed, DeviceSlice<T> out, void* stream = nullptr)
{
    const std::size_t n = detail::column_blocks<T>(width, packed.len, out.len, "unpack_column");
    detail::check(detail::Abi<T>::unpack((unsigned)width, packed.ptr, out.ptr, n, stream), "unpack_column");
}
// for b in blocks { T::unchecked_pack(width, &input[b*1024..], &mut packed[b*pl..]) }   (bitpacking.rs:76-96)
template <typename T>
inline void pack_column(std::size_t width, DeviceSlice<const T> input, DeviceSlice<T> packed, void* stream = nullptr)
{
    const std::size_t n = detail::column_blocks<T>(width, packed.len, input.len, "pack_column");
    detail::check(detail::Abi<T>::pack((unsigned)width, input.ptr, packed.ptr, n, stream), "pack_column");
}
// for b in blocks { T::undelta_pack::<W>(&packed[b], &bases[b], &mut out[b]) }   (delta.rs:47-63); bases: LANES per block
template <typename T>
inline void undelta_pack_column(std::size_t width, DeviceSlice<const T> packed, DeviceSlice<const T> bases, DeviceSlice<T> out,
                                void* stream = nullptr)
{
    const std::size_t n = detail::column_blocks<T>(width, packed.len, out.len, "undelta_pack_column");
    if (bases.len != n * (1024 / (sizeof(T) * 8))) throw std::length_error("undelta_pack_column: bases must hold LANES elements per block");
    detail::check(detail::Abi<T>::undelta_pack((unsigned)width, packed.ptr, bases.ptr, out.ptr, n, stream), "undelta_pack_column");
}
// for b in blocks { T::unfor_pack::<W>(&packed[b], references[b], &mut out[b]) }   (ffor.rs:38-50); one reference per block,
// or ONE for the whole column (references.len == 1)
template <typename T>
inline void unfor_pack_column(std::size_t width, DeviceSlice<const T> packed, DeviceSlice<const T> references, DeviceSlice<T> out,
                              void* stream = nullptr)
{
    const std::size_t n = detail::column_blocks<T>(width, packed.len, out.len, "unfor_pack_column");
    if (references.len != n && references.len != 1) throw std::length_error("unfor_pack_column: one reference per block, or one in all");
    detail::check(detail::Abi<T>::unfor_pack((unsigned)width, packed.ptr, references.ptr, references.len == 1 ? 0 : 1, out.ptr, n, stream),
                  "unfor_pack_column");
}

// The same loop over MANY SMALL ARRAYS ("chunks": Vortex keeps 64 Ki values per chunk), one launch: the four per-chunk device
// arrays of fl_<ty>_unpack_batch with their lengths.  unpack_chunks checks that the table is consistent (one entry per chunk in
// every array); what only the device can see -- each chunk's width, pointers' alignment, block count against max_blocks -- is
// checked by the kernel and reported through *err_flag (FL_DEVERR_*).
template <typename T> struct ChunkTable {
    DeviceSlice<const T* const> packed;          // [n_chunks] device pointers to the chunks' packed blocks
    DeviceSlice<T* const> out;                   // [n_chunks] device pointers to where each chunk decodes
    DeviceSlice<const std::uint8_t> widths;      // [n_chunks]
    DeviceSlice<const std::uint32_t> n_blocks;   // [n_chunks]
    std::uint32_t max_blocks = 0;                // host-side bound on n_blocks[c] (sizes the grid)
};
template <typename T>
inline void unpack_chunks(const ChunkTable<T>& t, std::uint32_t* d_err_flag = nullptr, void* stream = nullptr)
{
    const std::size_t n = t.widths.len;
    if (t.packed.len != n || t.out.len != n || t.n_blocks.len != n) throw std::length_error("unpack_chunks: every array of the table holds one entry per chunk");
    detail::check(detail::Abi<T>::unpack_batch(t.packed.ptr, t.out.ptr, t.widths.ptr, t.n_blocks.ptr, n, t.max_blocks, d_err_flag, stream), "unpack_chunks");
}
// A column whose blocks each have their own width: the caller loop
// `for b { T::unchecked_unpack(widths[b], ..) }` (bitpacking.rs:109-129) as one device call.
template <typename T> class MixedWidthPlan {
public:
    MixedWidthPlan(const std::uint8_t* widths, std::size_t n_blocks)
    { detail::check(fl_mixed_plan_create(sizeof(T) * 8, widths, n_blocks, &plan_), "fl_mixed_plan_create"); }
    ~MixedWidthPlan() { fl_mixed_plan_destroy(plan_); }
    MixedWidthPlan(const MixedWidthPlan&) = delete;
    MixedWidthPlan& operator=(const MixedWidthPlan&) = delete;
    std::size_t n_blocks() const { return fl_mixed_plan_n_blocks(plan_); }
    std::uint64_t packed_bytes() const { return fl_mixed_plan_packed_bytes(plan_); }
    void unpack_device(const T* d_packed, T* d_out, void* stream = nullptr) const
    { detail::check(detail::Abi<T>::unpack_mixed(plan_, d_packed, d_out, stream), "unpack_mixed"); }
    void pack_device(const T* d_in, T* d_packed, void* stream = nullptr) const
    { detail::check(detail::Abi<T>::pack_mixed(plan_, d_in, d_packed, stream), "pack_mixed"); }
    const std::uint8_t* widths_device() const { return fl_mixed_plan_widths(plan_); }
    const std::uint64_t* offsets_device() const { return fl_mixed_plan_offsets(plan_); }
private:
    fl_mixed_plan* plan_ = nullptr;
};

// The optional allocation helper of fastlanes_amd.h as an owner: a (packed, unpacked) buffer pair for a column of `packed_len` /
// `unpacked_len` ELEMENTS (either direction: in = what the op reads), placed by measurement by default (FL_LAYOUT_PROBE:
// synchronous, contents unspecified afterwards).  Never needed to use the codec -- every call above takes any aligned pointers.
template <typename T> class ColumnPair {
public:
    ColumnPair(std::size_t in_len, std::size_t out_len, std::size_t aux_len = 0, int layout = FL_LAYOUT_PROBE, void* stream = nullptr)
    {
        void *i = nullptr, *a = nullptr, *o = nullptr;
        detail::check(fl_column_pair_alloc(in_len * sizeof(T), aux_len * sizeof(T), out_len * sizeof(T), layout, stream, &i, &a, &o, &handle_,
                                           &layout_, probe_gbps_), "fl_column_pair_alloc");
        in_ = DeviceSlice<T>(static_cast<T*>(i), in_len);
        aux_ = DeviceSlice<T>(static_cast<T*>(a), aux_len);
        out_ = DeviceSlice<T>(static_cast<T*>(o), out_len);
    }
    ~ColumnPair() { fl_column_pair_free(handle_); }
    ColumnPair(const ColumnPair&) = delete;
    ColumnPair& operator=(const ColumnPair&) = delete;
    DeviceSlice<T> in() const { return in_; }
    DeviceSlice<T> aux() const { return aux_; }
    DeviceSlice<T> out() const { return out_; }
    int layout() const { return layout_; }                                  // FL_LAYOUT_SEPARATE / _ZONED / _INTERLEAVED: what was kept
    std::uint32_t probe_gbps(int layout) const { return layout >= 0 && layout < FL_LAYOUT_COUNT ? probe_gbps_[layout] : 0; }   // 0 = not measured
private:
    void* handle_ = nullptr;
    DeviceSlice<T> in_, aux_, out_;
    int layout_ = -1;
    std::uint32_t probe_gbps_[FL_LAYOUT_COUNT] = {0, 0, 0, 0};
};

// The same loop with the per-block widths / byte offsets already resident in HBM:
// nothing is built on the host.  A block with a width > T, a misaligned offset or bytes outside [0, packed_bytes) is skipped and its
// FL_DEVERR_* bit is ORed into *d_err_flag (device uint32, may be null).
template <typename T>
inline void unpack_widths_device(const std::uint8_t* d_widths, const std::uint64_t* d_offsets, const T* d_packed, std::size_t packed_bytes, T* d_out,
                                 std::size_t n_blocks, std::uint32_t* d_err_flag = nullptr, void* stream = nullptr)
{ detail::check(detail::Abi<T>::unpack_widths(d_widths, d_offsets, d_packed, packed_bytes, d_out, n_blocks, d_err_flag, stream), "unpack_widths"); }
template <typename T>
inline void pack_widths_device(const std::uint8_t* d_widths, const std::uint64_t* d_offsets, const T* d_in, T* d_packed, std::size_t packed_bytes,
                               std::size_t n_blocks, std::uint32_t* d_err_flag = nullptr, void* stream = nullptr)
{ detail::check(detail::Abi<T>::pack_widths(d_widths, d_offsets, d_in, d_packed, packed_bytes, n_blocks, d_err_flag, stream), "pack_widths"); }
template <typename T>
inline void unpack_single_widths_device(const std::uint8_t* d_widths, const std::uint64_t* d_offsets, const T* d_packed, std::size_t packed_bytes,
                                        std::size_t n_blocks, const std::uint64_t* d_indices, std::size_t n_indices, T* d_out,
                                        std::uint32_t* d_err_flag = nullptr, void* stream = nullptr)
{ detail::check(detail::Abi<T>::unpack_single_widths(d_widths, d_offsets, d_packed, packed_bytes, n_blocks, d_indices, n_indices, d_out, d_err_flag, stream), "unpack_single_widths"); }
// ... with FoR's / Delta's bodies: `for b { FoR::unfor_pack::<widths[b]>(.., d_references[b * reference_stride], ..) }` (ffor.rs:24-50),
// `for b { Delta::undelta_pack::<widths[b]>(.., &d_bases[b], ..) }` (delta.rs:47-63; `untranspose`: straight to original order), and the
// fused encode `pack::<widths[b]>(delta(transpose(..), &d_bases[b]))`.
template <typename T>
inline void unfor_pack_widths_device(const std::uint8_t* d_widths, const std::uint64_t* d_offsets, const T* d_packed, std::size_t packed_bytes,
                                     const T* d_references, std::size_t reference_stride, T* d_out, std::size_t n_blocks,
                                     std::uint32_t* d_err_flag = nullptr, void* stream = nullptr)
{ detail::check(detail::Abi<T>::unfor_pack_widths(d_widths, d_offsets, d_packed, packed_bytes, d_references, reference_stride, d_out, n_blocks, d_err_flag, stream), "unfor_pack_widths"); }
template <typename T>
inline void for_pack_widths_device(const std::uint8_t* d_widths, const std::uint64_t* d_offsets, const T* d_in, const T* d_references,
                                   std::size_t reference_stride, T* d_packed, std::size_t packed_bytes, std::size_t n_blocks,
                                   std::uint32_t* d_err_flag = nullptr, void* stream = nullptr)
{ detail::check(detail::Abi<T>::for_pack_widths(d_widths, d_offsets, d_in, d_references, reference_stride, d_packed, packed_bytes, n_blocks, d_err_flag, stream), "for_pack_widths"); }
template <typename T>
inline void undelta_pack_widths_device(const std::uint8_t* d_widths, const std::uint64_t* d_offsets, const T* d_packed, std::size_t packed_bytes,
                                       const T* d_bases, T* d_out, std::size_t n_blocks, bool untranspose = false,
                                       std::uint32_t* d_err_flag = nullptr, void* stream = nullptr)
{
    detail::check((untranspose ? detail::Abi<T>::undelta_pack_untranspose_widths : detail::Abi<T>::undelta_pack_widths)(
                      d_widths, d_offsets, d_packed, packed_bytes, d_bases, d_out, n_blocks, d_err_flag, stream), "undelta_pack_widths");
}
template <typename T>
inline void transpose_delta_pack_widths_device(const std::uint8_t* d_widths, const std::uint64_t* d_offsets, const T* d_in, const T* d_bases,
                                               T* d_packed, std::size_t packed_bytes, std::size_t n_blocks,
                                               std::uint32_t* d_err_flag = nullptr, void* stream = nullptr)
{ detail::check(detail::Abi<T>::transpose_delta_pack_widths(d_widths, d_offsets, d_in, d_bases, d_packed, packed_bytes, n_blocks, d_err_flag, stream), "transpose_delta_pack_widths"); }
// An encoder's width choice between block_min_max and for_pack_widths: widths[b] = bit length of maxs[b] - mins[b] (extension).
template <typename T>
inline void for_widths_device(const T* d_mins, const T* d_maxs, std::size_t n_blocks, std::uint8_t* d_widths, void* stream = nullptr)
{ detail::check(detail::Abi<T>::for_widths(d_mins, d_maxs, n_blocks, d_widths, stream), "for_widths"); }
template <typename T>
inline void widths_to_offsets_device(const std::uint8_t* d_widths, std::size_t n_blocks, std::uint64_t* d_offsets,
                                     std::uint64_t* d_total_bytes = nullptr, std::uint32_t* d_err_flag = nullptr, void* stream = nullptr)
{ detail::check(fl_widths_to_offsets(sizeof(T) * 8, d_widths, n_blocks, d_offsets, d_total_bytes, d_err_flag, stream), "widths_to_offsets"); }

// Many small arrays (a columnar engine's chunks) in one launch: device arrays of device pointers / widths / block counts.
template <typename T>
inline void unpack_batch_device(const T* const* d_packed, T* const* d_out, const std::uint8_t* d_widths, const std::uint32_t* d_n_blocks,
                                std::size_t n_arrays, std::uint32_t max_blocks, std::uint32_t* d_err_flag = nullptr, void* stream = nullptr)
{ detail::check(detail::Abi<T>::unpack_batch(d_packed, d_out, d_widths, d_n_blocks, n_arrays, max_blocks, d_err_flag, stream), "unpack_batch"); }
template <typename T>
inline void pack_batch_device(const T* const* d_in, T* const* d_packed, const std::uint8_t* d_widths, const std::uint32_t* d_n_blocks,
                              std::size_t n_arrays, std::uint32_t max_blocks, std::uint32_t* d_err_flag = nullptr, void* stream = nullptr)
{ detail::check(detail::Abi<T>::pack_batch(d_in, d_packed, d_widths, d_n_blocks, n_arrays, max_blocks, d_err_flag, stream), "pack_batch"); }

}  // namespace fastlanes
