// fused_for_pack.hip -- a USER-WRITTEN fused kernel on the PACK side of the device functor API
// (fastlanes_amd/csrc/fl_device.hpp: pack_rows), the counterpart of splicing a custom body into the reference's
// exported `pack!` macro (macros.rs:34-98) the way FoR::for_pack does (ffor.rs:24-36):
//
//     let reference = input.iter().min();                       // the encoder's frame of reference for this block
//     pack!(u32, W, output, lane, |$idx| { input[$idx].wrapping_sub(reference) });
//
// i.e. frame-of-reference encoding with the reference COMPUTED IN THE SAME KERNEL: the block is read once, its minimum is
// reduced while the rows sit in registers (over the thread's 32 rows x 4 lanes, then across the 8 column threads of the
// block), subtracted on the way into pack_rows' source functor, and written out next to the packed block.  With the library
// alone this takes block_min_max + for_pack: the unpacked block (128*T bytes) read twice.
// Specified as the oracle composition  for_pack::<W>(v, min(v))  and tested against it (tests/test_gpu_parity.py).
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I fastlanes_amd/csrc \
//               examples/fused_for_pack.hip -o examples/libfused_for_pack.so
#include "fl_kernels.hpp"

using namespace fl;

template <int W>
__global__ __launch_bounds__(WG) void k_min_for_pack_u32(StreamArgs a, uint32_t* __restrict__ mins)
{
    using T = uint32_t;
    constexpr int TB = Elem<T>::BITS;
    uint64_t tile;
    if (!tile_of_workgroup(a, tile)) return;
    const unsigned tid = threadIdx.x;
    const uint64_t blk = tile * BLOCKS_PER_WG + (tid >> 3);
    const unsigned c = tid & 7u;
    if (blk >= a.n_blocks) return;                     // whole 8-thread groups leave together
    // this thread's column of the unpacked block: all T rows, every load in flight
    const u32x4* un = a.in + blk * (uint64_t)Elem<T>::CELLS_PER_BLOCK + c;
    Cell<T> rows[TB];
    iterate_rows<T>([&](auto R, auto CELL) { rows[decltype(R)::value] = load_cell<T, true>(un + decltype(CELL)::value); });
    // the block's minimum: over this thread's rows and lanes, then over the 8 column threads of the block
    T mn = ~(T)0;
    static_for<TB>([&](auto R) {
        for (int e = 0; e < 4; ++e) mn = rows[decltype(R)::value].x[e] < mn ? rows[decltype(R)::value].x[e] : mn;
    });
    for (int m = 1; m < 8; m <<= 1) {
        const T o = (T)__shfl_xor(mn, m, 8);
        mn = o < mn ? o : mn;
    }
    if (c == 0) mins[blk] = mn;
    const Cell<T> ref = Cell<T>::splat(mn);
    // the spliced body: pack_rows asks for row R's cell (any order it likes) and hands back finished packed word-rows
    const TileStore<(W ? W : 1) * 128> st(a.out, tile, a.n_blocks, tid);
    pack_rows<T, W>([&](auto R) { return rows[decltype(R)::value].sub(ref); },                  // ffor.rs:32-34
                    [&](auto Wd, const Cell<T>& word) { st.store(8 * decltype(Wd)::value, word); });
}

// C entry point: d_in [n_blocks * 1024] u32 -> d_packed [n_blocks * 32 * width] u32, d_mins [n_blocks]; width 0..32
extern "C" int example_min_for_pack_u32(unsigned width, const uint32_t* d_in, uint32_t* d_packed, uint32_t* d_mins, size_t n_blocks,
                                        void* stream)
{
    if (width > 32) return 1;                          // FL_ERR_WIDTH
    if (n_blocks == 0) return 0;
    StreamArgs a;
    a.in = reinterpret_cast<const u32x4*>(d_in);
    a.out = reinterpret_cast<u32x4*>(d_packed);
    a.aux = nullptr;
    a.aux_stride = 0;
    a.n_blocks = n_blocks;
    const unsigned grid = plan_grid(a, WIN_PACK, 32);
    hipStream_t s = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    switch (width) {
#define CASE(W) case W: hipLaunchKernelGGL((k_min_for_pack_u32<W>), dim3(grid), dim3(WG), 0, s, a, d_mins); break;
    CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16)
    CASE(17) CASE(18) CASE(19) CASE(20) CASE(21) CASE(22) CASE(23) CASE(24) CASE(25) CASE(26) CASE(27) CASE(28) CASE(29) CASE(30) CASE(31) CASE(32)
#undef CASE
    }
    return (int)hipGetLastError();
}
