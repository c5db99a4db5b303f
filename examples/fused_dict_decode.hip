// fused_dict_decode.hip -- a USER-WRITTEN fused kernel on the device functor API
// (fastlanes_amd/csrc/fl_device.hpp), the counterpart of splicing a custom body into the
// reference's exported `unpack!` macro (macros.rs:100-174):
//
//     unpack!(u32, W, packed, lane, |$idx, $elem| { output[$idx] = dict[$elem as usize]; });
//
// i.e. dictionary decoding fused into bit-unpacking: codes are never materialised.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I fastlanes_amd/csrc \
//               examples/fused_dict_decode.hip -o examples/libfused_dict_decode.so
#include "fl_kernels.hpp"

using namespace fl;

template <int W>
__global__ __launch_bounds__(WG) void k_dict_unpack_u32(StreamArgs a, const uint32_t* __restrict__ dict)
{
    using T = uint32_t;
    uint64_t tile;
    if (!tile_of_workgroup(a, tile)) return;
    const unsigned tid = threadIdx.x;
    const uint64_t blk = tile * BLOCKS_PER_WG + (tid >> 3);
    const unsigned c = tid & 7u;
    if (blk >= a.n_blocks) return;
    Cell<T> in[W];
    const u32x4* pk = a.in + blk * (uint64_t)(8 * W) + c;
    static_for<W>([&](auto Wd) { in[decltype(Wd)::value] = load_cell<T>(pk + 8 * decltype(Wd)::value); });
    const TileStore<Elem<T>::CELLS_PER_BLOCK * 16> st(a.out, tile, a.n_blocks, tid);
    // the spliced body: called once per row, in row order, with this column's 4 codes
    unpack_rows<T, W>(in, [&](auto R, const Cell<T>& codes) {
        Cell<T> vals;
        for (int i = 0; i < 4; ++i) vals.x[i] = dict[codes.x[i]];
        st.store(Elem<T>::row_cell(decltype(R)::value), vals);
    });
}

// C entry point: width-8 codes (dictionary of up to 256 u32 values)
extern "C" int example_dict_unpack_u32_w8(const uint32_t* d_packed, const uint32_t* d_dict, uint32_t* d_out,
                                          size_t n_blocks, void* stream)
{
    if (n_blocks == 0) return 0;
    StreamArgs a;
    a.in = reinterpret_cast<const u32x4*>(d_packed);
    a.out = reinterpret_cast<u32x4*>(d_out);
    a.aux = nullptr;
    a.aux_stride = 0;
    a.n_blocks = n_blocks;
    const unsigned grid = plan_grid(a);
    hipLaunchKernelGGL((k_dict_unpack_u32<8>), dim3(grid), dim3(WG), 0, static_cast<hipStream_t>(stream), a, d_dict);
    return (int)hipGetLastError();
}
