// fl_inst.hip -- explicit instantiation unit.  Compiled once per
// (element type, family) with -DFL_T=<type> -DFL_FAMILY=<n> so the 124 (T,W)
// pairs x 5 width-parameterised kernels build in parallel:
//   0 unpack (store)   1 unfor_pack   2 undelta_pack   3 pack   4 for_pack
//   5 delta / undelta / transpose / untranspose / unpack_single
//   6 wave-per-block kernels: unpack / pack over per-block (or uniform) widths (fl_widths.hpp), Delta chains (fl_chain.hpp)
//   10 fused consumers: unpack_block_sums, block_min_max   11 / 12 unpack_compare (selection masks: x <= k / x == k)
//   8 undelta_pack+untranspose (fused decode to original order)   9 transpose+delta+pack (fused encode)
#include "fl_kernels.hpp"
#include "fl_misc.hpp"
#include "fl_widths.hpp"
#include "fl_chain.hpp"
#include "fl_batch.hpp"
#include "fl_consume.hpp"

namespace fl {
using T = FL_T;
using Ws = std::make_integer_sequence<int, Elem<T>::BITS + 1>;

#if FL_FAMILY == 0
static constexpr WidthTable<T> t_store = make_unpack_table<T, BODY_STORE>(Ws{});
template <> const WidthTable<T>& unpack_table_impl<T, BODY_STORE>() { return t_store; }
#elif FL_FAMILY == 1
static constexpr WidthTable<T> t_addref = make_unpack_table<T, BODY_ADD_REF>(Ws{});
template <> const WidthTable<T>& unpack_table_impl<T, BODY_ADD_REF>() { return t_addref; }
#elif FL_FAMILY == 2
static constexpr WidthTable<T> t_undelta = make_unpack_table<T, BODY_UNDELTA>(Ws{});
template <> const WidthTable<T>& unpack_table_impl<T, BODY_UNDELTA>() { return t_undelta; }
#elif FL_FAMILY == 3
static constexpr WidthTable<T> t_pack = make_pack_table<T, PACK_PLAIN>(Ws{});
template <> const WidthTable<T>& pack_table_impl<T, PACK_PLAIN>() { return t_pack; }
#elif FL_FAMILY == 4
static constexpr WidthTable<T> t_forpack = make_pack_table<T, PACK_FOR>(Ws{});
template <> const WidthTable<T>& pack_table_impl<T, PACK_FOR>() { return t_forpack; }
#elif FL_FAMILY == 5
template <> stream_launch_t delta_launcher<T>(bool inverse)
{
    return inverse ? &launch_delta<T, true> : &launch_delta<T, false>;
}
template <> stream_launch_t transpose_launcher<T>(bool inverse)
{
    return inverse ? &launch_transpose<T, true> : &launch_transpose<T, false>;
}
template <> hipError_t unpack_single_launch<T>(const SingleArgs& a, hipStream_t s)
{
    return launch_unpack_single<T>(a, s);
}
#elif FL_FAMILY == 6
template <> widths_launch_t widths_launcher<T>(bool pack)
{
    return pack ? &launch_widths<T, true> : &launch_widths<T, false>;
}
template <> batch_launch_t batch_launcher<T>(bool pack)
{
    return pack ? &launch_batch<T, true> : &launch_batch<T, false>;
}
template <> batch_launch_t batch_chain_launcher<T>(int op)
{
    switch (op) {
    case OP_UNDELTA_PACK: return &launch_batch_chain<T, SRC_PACKED, CHAIN_UNDELTA, SNK_ROWS>;
    case OP_UNDELTA_PACK_UNTRANSPOSE: return &launch_batch_chain<T, SRC_PACKED, CHAIN_UNDELTA, SNK_ORIGINAL>;
    case OP_TRANSPOSE_DELTA_PACK: return &launch_batch_chain<T, SRC_ORIGINAL, CHAIN_DELTA, SNK_PACKED>;
    default: return nullptr;
    }
}
template <> chain_launch_t chain_launcher<T>(int op)
{
    switch (op) {
    case OP_UNDELTA_PACK: return &launch_chain<T, SRC_PACKED, CHAIN_UNDELTA, SNK_ROWS, RD_AUTO>;
    case OP_UNDELTA: return &launch_chain<T, SRC_ROWS, CHAIN_UNDELTA, SNK_ROWS>;
    case OP_DELTA: return &launch_chain<T, SRC_ROWS, CHAIN_DELTA, SNK_ROWS>;
    case OP_UNTRANSPOSE: return &launch_chain<T, SRC_ROWS, CHAIN_NONE, SNK_ORIGINAL>;
    case OP_TRANSPOSE: return &launch_chain<T, SRC_ORIGINAL, CHAIN_NONE, SNK_ROWS>;
    case OP_UNDELTA_PACK_UNTRANSPOSE: return &launch_chain<T, SRC_PACKED, CHAIN_UNDELTA, SNK_ORIGINAL, RD_AUTO>;
    case OP_TRANSPOSE_DELTA_PACK: return &launch_chain<T, SRC_ORIGINAL, CHAIN_DELTA, SNK_PACKED>;
    default: return nullptr;
    }
}
template <> chain_launch_t chain_launcher_two_blocks<T>(int op)
{
    if constexpr (sizeof(T) >= 4) {
        if (op == OP_UNDELTA_PACK) return &launch_chain<T, SRC_PACKED, CHAIN_UNDELTA, SNK_ROWS, RD_VGPR, 2>;
    }
    return nullptr;
}
template <> chain_launch_t chain_widths_launcher<T>(int op)
{
    constexpr unsigned B = chain_blocks_per_wave<T>();
    if constexpr (sizeof(T) == 1) {                         // u8's decode: column lanes, pipelined (fl_chain.hpp)
        if (op == OP_UNDELTA_PACK) return &launch_chain_columns_pipelined<T, SNK_ROWS>;
        if (op == OP_UNDELTA_PACK_UNTRANSPOSE) return &launch_chain_columns_pipelined<T, SNK_ORIGINAL>;
        if (op == OP_TRANSPOSE_DELTA_PACK) return &launch_chain_columns_encode_pipelined<T>;
    }
    switch (op) {
    // (one block per wavefront -- u32 / u64 -- goes through chain_stage_source: RD_AUTO = a mixed-width column's packed rows stream
    // non-temporally by LDS-DMA, as in unpack_widths; the lockstep form of the narrow types always did)
    case OP_UNDELTA_PACK: return &launch_chain<T, SRC_PACKED, CHAIN_UNDELTA, SNK_ROWS, RD_AUTO, B>;
    case OP_UNDELTA_PACK_UNTRANSPOSE: return &launch_chain<T, SRC_PACKED, CHAIN_UNDELTA, SNK_ORIGINAL, RD_AUTO, B>;
    case OP_TRANSPOSE_DELTA_PACK: return &launch_chain<T, SRC_ORIGINAL, CHAIN_DELTA, SNK_PACKED, RD_VGPR, B>;
    default: return nullptr;
    }
}
#elif FL_FAMILY == 8
static constexpr WidthTable<T> t_undelta_untr = make_unpack_table<T, BODY_UNDELTA_UNTRANSPOSE>(Ws{});
template <> const WidthTable<T>& unpack_table_impl<T, BODY_UNDELTA_UNTRANSPOSE>() { return t_undelta_untr; }
#elif FL_FAMILY == 9
static constexpr WidthTable<T> t_tr_delta_pack = make_pack_table<T, PACK_TRANSPOSE_DELTA>(Ws{});
template <> const WidthTable<T>& pack_table_impl<T, PACK_TRANSPOSE_DELTA>() { return t_tr_delta_pack; }
#elif FL_FAMILY == 10
static constexpr ReduceTable<T> t_sums = make_sum_table<T>(Ws{});
template <> const ReduceTable<T>& sum_table_impl<T>() { return t_sums; }
template <> reduce_launch_t min_max_launcher<T>() { return &launch_block_min_max<T>; }
#elif FL_FAMILY == 11
static constexpr CompareTable<T> t_compare_le = make_compare_table<T, false>(Ws{});
template <> const CompareTable<T>& compare_table_impl<T, false>() { return t_compare_le; }
#elif FL_FAMILY == 12
static constexpr CompareTable<T> t_compare_eq = make_compare_table<T, true>(Ws{});
template <> const CompareTable<T>& compare_table_impl<T, true>() { return t_compare_eq; }
#else
#error "FL_FAMILY must be 0..6 or 8..12"
#endif
}  // namespace fl
