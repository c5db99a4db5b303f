// fl_batch.hpp -- many small arrays in ONE launch.  The callers SURVEY.md 8(b) describes (Vortex) hold a column as chunks of
// 64 Ki values (64 blocks) and loop
//     for chunk in chunks { for b in 0..chunk.blocks { T::unchecked_unpack(chunk.width, &chunk.packed[b*..], &mut chunk.out[b*1024..]) } }
// (bitpacking.rs:109-129 inside the loop shape of benches/bitpacking.rs:80-97).  One device-tier call per chunk is
// launch-bound (3 us per launch = 20 G ints/s at 64 blocks); fl_<ty>_unpack_widths needs the chunks in one allocation.  Here
// the chunks are given as DEVICE ARRAYS OF POINTERS: array a has n_blocks[a] blocks of width widths[a] at packed[a] and
// decodes to out[a].  Mapping: one wavefront per block (or per `bpw` consecutive blocks, all requested up front), workgroup
// (array, 4 * bpw consecutive blocks of it); the grid is sized from the caller's bound max_blocks >= n_blocks[a], workgroups
// past an array's end leave at once.  The block kernel is the runtime-width wave-per-block one (fl_widths.hpp), fed a
// per-array argument block.
#pragma once
#include "fl_chain.hpp"

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
    unsigned window_shift;       // tile-map window (fl_kernels.hpp: xcd_tile)
    unsigned tiles_per_array;    // ceil(max_blocks / (4 * bpw))
    unsigned max_blocks;         // the caller's bound on n_blocks[a]
    unsigned bpw;                // consecutive blocks of the array per wavefront (>= 1); a workgroup takes 4 * bpw
    unsigned prefetch;           // bpw > 1: all of a wavefront's blocks are requested up front by LDS-DMA (one image per block)
    const char* const* bases;    // Delta's entries only: [n_arrays] device pointers to the arrays' bases, [n_blocks[a]][LANES] each
};

// Array `arr`'s descriptor -- block count, width, the two pointers -- as four INDEPENDENT loads through the vector memory
// path, in flight together, ONE wait, then broadcast to SGPRs.  (Round 3 read n_blocks[arr] first, tested it, and only then
// fetched the pointers and the width with scalar loads: three dependent memory round trips in front of a workgroup whose
// whole job is 4 blocks; the batch ran 16 % behind the same blocks as one contiguous column, profiles/r03_sweep_batch.txt.)
struct ArrayDesc { unsigned n_blocks, width; uint64_t packed, unpacked; };
__device__ __forceinline__ ArrayDesc array_desc(const BatchArgs& b, unsigned arr)
{
    const unsigned i = arr + opaque_zero();
    const unsigned nb = b.n_blocks[i];
    const unsigned w = b.widths[i];
    const uint64_t pk = reinterpret_cast<const uint64_t*>(b.packed)[i];
    const uint64_t un = reinterpret_cast<const uint64_t*>(b.unpacked)[i];
    ArrayDesc d;
    d.n_blocks = (unsigned)__builtin_amdgcn_readfirstlane(nb);
    d.width = (unsigned)__builtin_amdgcn_readfirstlane(w);
    d.packed = wave_uniform_u64(pk);
    d.unpacked = wave_uniform_u64(un);
    return d;
}

template <typename T, bool PACK>
__global__ __launch_bounds__(WG) void k_batch(BatchArgs b)
{
    using G = WaveBlock<T>;
    extern __shared__ __attribute__((aligned(16))) char lds_all[];
    const uint64_t n_tiles = b.n_arrays * b.tiles_per_array;
    const uint64_t tile = xcd_tile(blockIdx.x, b.tiles_per_xcd, b.window_shift);
    if (tile >= n_tiles) return;
    // the launcher keeps the grid below 2^31 workgroups: 32-bit division (a 64-bit one is ~3x the scalar instructions)
    const unsigned arr = (unsigned)tile / b.tiles_per_array;
    const unsigned tile_in_arr = (unsigned)tile - arr * b.tiles_per_array;
    const unsigned tid = threadIdx.x;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
    const unsigned first = (tile_in_arr * (WG / 64) + wave) * b.bpw;
    const ArrayDesc d = array_desc(b, arr);
    if (first >= d.n_blocks) return;
    // a bound that is too small would leave the array's tail undecoded without a trace: flag it (once per array)
    if (first == 0 && d.n_blocks > b.max_blocks) raise_device_error(b.err_flag, DEVERR_BOUNDS, lane);
    WidthsArgs a;
    a.packed = reinterpret_cast<const char*>(d.packed);
    a.unpacked = reinterpret_cast<char*>(d.unpacked);
    a.widths = nullptr;
    a.offsets = nullptr;
    a.err_flag = b.err_flag;
    a.refs = b.refs ? static_cast<const char*>(b.refs) + (uint64_t)arr * sizeof(T) : nullptr;   // every block of the array shares it
    a.ref_stride = 0;
    a.n_blocks = d.n_blocks;
    a.tiles_per_xcd = 0;
    a.window_shift = 63;
    a.uniform_width = d.width;
    a.bpw = b.bpw;
    a.packed_bytes = 0;
    a.prefetch = b.prefetch;
    a.linear_map = 0;
    a.nt_from = (unsigned)WaveBlock<T>::TB / 2u;
    // per-array preconditions the host cannot check (the pointers live in HBM): 16-byte alignment; the width check
    // (bitpacking.rs:93,126) is the block kernel's
    if (((d.packed | d.unpacked) & 15u) != 0 || !d.unpacked || (!d.packed && d.width != 0)) {
        raise_device_error(b.err_flag, DEVERR_ALIGN, lane);
        return;
    }
    const unsigned left = d.n_blocks - first;
    const unsigned count = left < b.bpw ? left : b.bpw;
    char* lds = lds_all + wave * G::BLOCK_BYTES * (b.prefetch ? b.bpw : 1u);
    if (b.prefetch && count > 1) {
        if constexpr (PACK) pack_blocks_wave_prefetched<T>(a, first, count, lds, lane);
        else unpack_blocks_wave_prefetched<T>(a, first, count, lds, lane);
        return;
    }
    for (unsigned j = 0; j < count; ++j) {
        if constexpr (PACK) pack_block_wave<T, RD_VGPR>(a, first + j, lds, lane);
        else unpack_block_wave<T, RD_AUTO>(a, first + j, lds, lane);
    }
}

// Delta over many small arrays: `for b in 0..chunk.blocks { Delta::undelta_pack::<W>(&chunk.packed[b*..], &chunk.bases[b], ..) }`
// (delta.rs:47-63) -- and the two fused transpose extensions -- with the pipeline kernel's stages (fl_chain.hpp), one wavefront per
// block, the per-array argument block built from the descriptor (five independent loads: the bases pointer rides along).
// blocks per wavefront: the narrow types keep several in flight (fl_chain.hpp: chain_blocks_lockstep) -- except u16's encode, whose
// unpacked blocks are better read through VGPRs than by LDS-DMA (profiles/abchain_narrow_r04.txt)
template <typename T, int SRC> constexpr unsigned batch_chain_blocks_per_wave()
{
    return sizeof(T) == 1 ? 4u : (sizeof(T) == 2 && SRC == SRC_PACKED) ? 2u : 1u;
}

template <typename T, int SRC, int BODY, int SNK>
__global__ __launch_bounds__(WG) void k_batch_chain(BatchArgs b)
{
    extern __shared__ __attribute__((aligned(16))) char lds_all[];
    constexpr unsigned BPW = batch_chain_blocks_per_wave<T, SRC>();
    constexpr unsigned WAVE_LDS = chain_wave_lds<T, SRC, SNK>() * BPW;
    const uint64_t n_tiles = b.n_arrays * b.tiles_per_array;
    const uint64_t tile = xcd_tile(blockIdx.x, b.tiles_per_xcd, b.window_shift);
    if (tile >= n_tiles) return;
    const unsigned arr = (unsigned)tile / b.tiles_per_array;
    const unsigned tile_in_arr = (unsigned)tile - arr * b.tiles_per_array;
    const unsigned tid = threadIdx.x;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
    const unsigned blk = (tile_in_arr * (WG / 64) + wave) * BPW;
    const uint64_t bases_v = reinterpret_cast<const uint64_t*>(b.bases)[arr + opaque_zero()];   // in flight with the descriptor's loads
    const ArrayDesc d = array_desc(b, arr);
    const uint64_t bases = wave_uniform_u64(bases_v);
    if (blk >= d.n_blocks) return;
    if (blk == 0 && d.n_blocks > b.max_blocks) raise_device_error(b.err_flag, DEVERR_BOUNDS, lane);
    if (d.width > (unsigned)WaveBlock<T>::TB) {                                // delta.rs:11-16: W <= T is a type-level bound there
        raise_device_error(b.err_flag, DEVERR_WIDTH, lane);
        return;
    }
    if (((d.packed | d.unpacked | bases) & 15u) != 0 || !d.unpacked || !bases || (!d.packed && d.width != 0)) {
        raise_device_error(b.err_flag, DEVERR_ALIGN, lane);
        return;
    }
    ChainArgs a;
    a.in = reinterpret_cast<const char*>(SRC == SRC_PACKED ? d.packed : d.unpacked);
    a.out = reinterpret_cast<char*>(SNK == SNK_PACKED ? d.packed : d.unpacked);
    a.bases = reinterpret_cast<const char*>(bases);
    a.n_blocks = d.n_blocks;
    a.tiles_per_xcd = 0;
    a.window_shift = 63;
    a.width = d.width;
    a.widths = nullptr;
    a.offsets = nullptr;
    a.err_flag = b.err_flag;
    a.packed_bytes = 0;
    a.nt_from = (unsigned)WaveBlock<T>::TB / 2u;
    if constexpr (BPW == 1) {
        chain_one_block<T, SRC, BODY, SNK, RD_VGPR>(a, blk, lds_all + wave * WAVE_LDS, lane);
    } else {
        const unsigned left = d.n_blocks - blk;
        chain_blocks_lockstep<T, SRC, BODY, SNK, BPW>(a, blk, left < BPW ? left : BPW, lds_all + wave * WAVE_LDS, lane);
    }
}

template <typename T, int SRC, int BODY, int SNK>
hipError_t launch_batch_chain(const BatchArgs& b0, uint32_t max_blocks, int waves, hipStream_t s)
{
    if (b0.n_arrays == 0 || max_blocks == 0) return hipSuccess;
    BatchArgs b = b0;
    constexpr unsigned TILE_BLOCKS = batch_chain_blocks_per_wave<T, SRC>() * (WG / 64);
    b.bpw = batch_chain_blocks_per_wave<T, SRC>();
    b.prefetch = 0;
    b.tiles_per_array = (unsigned)(((uint64_t)max_blocks + TILE_BLOCKS - 1) / TILE_BLOCKS);
    b.max_blocks = max_blocks;
    const uint64_t n_tiles = b.n_arrays * b.tiles_per_array;
    b.tiles_per_xcd = (n_tiles + 7) / 8;
    if (b.tiles_per_array == 0 || b.tiles_per_xcd * 8 > 0x7fffffffull) return hipErrorInvalidValue;
    b.window_shift = tile_window_shift(chain_window_op<SRC, BODY, SNK>(), WaveBlock<T>::TB, TILE_BLOCKS);
    const unsigned need = TILE_BLOCKS * chain_wave_lds<T, SRC, SNK>();
    if (waves < 3) waves = 3;
    const unsigned pad = (CU_LDS_BYTES * (unsigned)WG / ((unsigned)waves * 256u)) & ~1023u;
    FL_LAUNCH((k_batch_chain<T, SRC, BODY, SNK>), dim3((unsigned)(b.tiles_per_xcd * 8)), dim3(WG), pad > need ? pad : need, s, b);
    return hipGetLastError();
}

typedef hipError_t (*batch_launch_t)(const BatchArgs&, uint32_t max_blocks, int waves, hipStream_t);
// op: OP_UNDELTA_PACK / OP_UNDELTA_PACK_UNTRANSPOSE / OP_TRANSPOSE_DELTA_PACK (fl_chain.hpp)
template <typename T> batch_launch_t batch_chain_launcher(int op);

template <typename T, bool PACK>
hipError_t launch_batch(const BatchArgs& b0, uint32_t max_blocks, int waves, hipStream_t s)
{
    if (b0.n_arrays == 0 || max_blocks == 0) return hipSuccess;
    BatchArgs b = b0;
    if (b.bpw == 0) b.bpw = 1;
    if (b.bpw < 2 || b.bpw > 16) b.prefetch = 0;
    if (b.prefetch && (WG / 64) * b.bpw * WaveBlock<T>::BLOCK_BYTES > 64u * 1024u) b.prefetch = 0;   // images would not fit a workgroup's LDS
    const uint64_t tile_blocks = (uint64_t)b.bpw * (WG / 64);
    b.tiles_per_array = (unsigned)(((uint64_t)max_blocks + tile_blocks - 1) / tile_blocks);            // 64-bit: no wrap near 2^32
    b.max_blocks = max_blocks;
    const uint64_t n_tiles = b.n_arrays * b.tiles_per_array;
    b.tiles_per_xcd = (n_tiles + 7) / 8;
    if (b.tiles_per_array == 0 || b.tiles_per_xcd * 8 > 0x7fffffffull) return hipErrorInvalidValue;
    b.window_shift = tile_window_shift(PACK ? WIN_PACK : WIN_UNPACK, WaveBlock<T>::TB, (unsigned)tile_blocks);
    const unsigned lds = widths_lds_bytes<T>(waves, b.prefetch ? b.bpw : 1u);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    FL_LAUNCH((k_batch<T, PACK>), dim3((unsigned)(b.tiles_per_xcd * 8)), dim3(WG), lds, s, b);
    return hipGetLastError();
}

template <typename T> batch_launch_t batch_launcher(bool pack);

}  // namespace fl
