// fl_batch.hpp -- many small arrays in ONE launch.  The callers SURVEY.md 8(b) describes (Vortex) hold a column as chunks of
// 64 Ki values (64 blocks) and loop
//     for chunk in chunks { for b in 0..chunk.blocks { T::unchecked_unpack(chunk.width, &chunk.packed[b*..], &mut chunk.out[b*1024..]) } }
// (bitpacking.rs:109-129 inside the loop shape of benches/bitpacking.rs:80-97).  One device-tier call per chunk is
// launch-bound (3 us per launch = 20 G ints/s at 64 blocks); fl_<ty>_unpack_widths needs the chunks in one allocation.  Here
// the chunks are given as DEVICE ARRAYS OF POINTERS: array a has n_blocks[a] blocks of width widths[a] at packed[a] and
// decodes to out[a].  Mapping: one wavefront per block, workgroup (array, 4 consecutive blocks of it); the grid is sized
// from the caller's bound max_blocks >= n_blocks[a], workgroups past an array's end leave at once.  The block kernel is
// the runtime-width wave-per-block one (fl_widths.hpp), fed a per-array argument block.
#pragma once
#include "fl_widths.hpp"

namespace fl {

struct BatchArgs {
    const char* const* packed;   // [n_arrays] device pointers to the packed arrays
    char* const* unpacked;       // [n_arrays] device pointers to the unpacked arrays
    const uint8_t* widths;       // [n_arrays]
    const uint32_t* n_blocks;    // [n_arrays]
    uint32_t* err_flag;          // FL_DEVERR_* bits; may be nullptr
    const void* refs;            // FoR: references[a], one per ARRAY (ffor.rs:24-50); nullptr = plain BitPacking
    uint64_t n_arrays;
    uint64_t tiles_per_xcd;
    unsigned tiles_per_array;    // ceil(max_blocks / 4)
    unsigned max_blocks;         // the caller's bound on n_blocks[a]
};

template <typename T, bool PACK>
__global__ __launch_bounds__(WG) void k_batch(BatchArgs b)
{
    using G = WaveBlock<T>;
    extern __shared__ __attribute__((aligned(16))) char lds_all[];
    const uint64_t n_tiles = b.n_arrays * b.tiles_per_array;
    const uint64_t tile = (uint64_t)(blockIdx.x & 7u) * b.tiles_per_xcd + (blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    const uint64_t arr = tile / b.tiles_per_array;
    const unsigned tid = threadIdx.x;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
    const uint64_t blk = (tile - arr * b.tiles_per_array) * (WG / 64) + wave;
    const unsigned nb = b.n_blocks[arr];
    if (blk >= nb) return;
    // a bound that is too small would leave the array's tail undecoded without a trace: flag it (once per array)
    if (blk == 0 && nb > b.max_blocks) raise_device_error(b.err_flag, DEVERR_BOUNDS, lane);
    WidthsArgs a;
    a.packed = b.packed[arr];
    a.unpacked = b.unpacked[arr];
    a.widths = nullptr;
    a.offsets = nullptr;
    a.err_flag = b.err_flag;
    a.refs = b.refs ? static_cast<const char*>(b.refs) + arr * sizeof(T) : nullptr;   // every block of the array shares it
    a.ref_stride = 0;
    a.n_blocks = nb;
    a.tiles_per_xcd = 0;
    a.uniform_width = b.widths[arr];
    a.bpw = 1;
    a.packed_bytes = 0;
    a.prefetch = 0;
    a.linear_map = 0;
    // per-array preconditions the host cannot check (the pointers live in HBM): 16-byte alignment; the width check
    // (bitpacking.rs:93,126) is the block kernel's
    if (((reinterpret_cast<uintptr_t>(a.packed) | reinterpret_cast<uintptr_t>(a.unpacked)) & 15u) != 0 ||
        !a.unpacked || (!a.packed && a.uniform_width != 0)) {
        raise_device_error(b.err_flag, DEVERR_ALIGN, lane);
        return;
    }
    char* lds = lds_all + wave * G::BLOCK_BYTES;
    if constexpr (PACK) pack_block_wave<T, RD_VGPR>(a, blk, lds, lane);
    else unpack_block_wave<T, RD_AUTO>(a, blk, lds, lane);
}

typedef hipError_t (*batch_launch_t)(const BatchArgs&, uint32_t max_blocks, int waves, hipStream_t);

template <typename T, bool PACK>
hipError_t launch_batch(const BatchArgs& b0, uint32_t max_blocks, int waves, hipStream_t s)
{
    if (b0.n_arrays == 0 || max_blocks == 0) return hipSuccess;
    BatchArgs b = b0;
    b.tiles_per_array = (max_blocks + (WG / 64) - 1) / (WG / 64);
    b.max_blocks = max_blocks;
    const uint64_t n_tiles = b.n_arrays * b.tiles_per_array;
    b.tiles_per_xcd = (n_tiles + 7) / 8;
    if (b.tiles_per_xcd * 8 > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_batch<T, PACK>), dim3((unsigned)(b.tiles_per_xcd * 8)), dim3(WG), widths_lds_bytes<T>(waves), s, b);
    return hipGetLastError();
}

template <typename T> batch_launch_t batch_launcher(bool pack);

}  // namespace fl
