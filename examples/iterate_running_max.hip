// iterate_running_max.hip -- a USER-WRITTEN kernel on fl::iterate_rows (fastlanes_amd/csrc/fl_device.hpp), the
// counterpart of splicing a stateful body into the reference's exported `iterate!` macro (macros.rs:11-32), exactly
// as Delta::undelta does with a running sum (delta.rs:36-45) -- here with a running MAXIMUM per FastLanes lane:
//
//     for lane in 0..LANES { let mut run = base[lane];
//         iterate!(u32, lane, |$idx| { run = max(run, input[$idx]); output[$idx] = run; }); }
//
// (a building block of "frame of running max" / monotone-envelope encodings over transposed data).
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I fastlanes_amd/csrc \
//               examples/iterate_running_max.hip -o examples/libiterate_running_max.so
#include "fl_kernels.hpp"

using namespace fl;

__global__ __launch_bounds__(WG) void k_running_max_u32(StreamArgs a)
{
    using T = uint32_t;
    uint64_t tile;
    if (!tile_of_workgroup(a, tile)) return;
    const unsigned tid = threadIdx.x;
    const uint64_t blk = tile * BLOCKS_PER_WG + (tid >> 3);
    const unsigned c = tid & 7u;
    if (blk >= a.n_blocks) return;
    const u32x4* in = a.in + blk * (uint64_t)Elem<T>::CELLS_PER_BLOCK + c;
    Cell<T> run = load_cell<T>(static_cast<const u32x4*>(a.aux) + blk * 8 + c);     // base[lane], 4 lanes per cell
    const TileStore<Elem<T>::CELLS_PER_BLOCK * 16> st(a.out, tile, a.n_blocks, tid);
    // the spliced body: once per row, in row order; CELL is where index(row, lane) lives for this column
    iterate_rows<T>([&](auto, auto CELL) {
        constexpr int cell = decltype(CELL)::value;
        const Cell<T> v = load_cell<T>(in + cell);
        for (int i = 0; i < 4; ++i) run.x[i] = run.x[i] > v.x[i] ? run.x[i] : v.x[i];
        st.store(cell, run);
    });
}

extern "C" int example_running_max_u32(const uint32_t* d_in, const uint32_t* d_base, uint32_t* d_out, size_t n_blocks,
                                       void* stream)
{
    if (n_blocks == 0) return 0;
    StreamArgs a;
    a.in = reinterpret_cast<const u32x4*>(d_in);
    a.out = reinterpret_cast<u32x4*>(d_out);
    a.aux = d_base;
    a.aux_stride = 0;
    a.n_blocks = n_blocks;
    const unsigned grid = plan_grid(a);
    hipLaunchKernelGGL(k_running_max_u32, dim3(grid), dim3(WG), 0, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}
