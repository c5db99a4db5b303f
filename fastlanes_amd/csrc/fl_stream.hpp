// fl_stream.hpp -- a BARE STREAM with the wave-per-block kernels' access shape and nothing else: the yardstick every streaming
// kernel of the library is read against (bench.py: roofline.bare_stream_GBps / frac_of_bare_stream; the layout probe of
// fl_column_pair_alloc).  One wavefront per UNIT (a 1024-value block's worth of bytes): in_unit bytes read as 1-KiB contiguous
// loads (all in flight; through a descriptor that ends at in_unit), aux_unit bytes likewise (Delta's bases), XORed, out_unit
// bytes written as 1-KiB contiguous `sc1 nt` stores; the XCD-contiguous tile map of fl_kernels.hpp (4 units per workgroup),
// occupancy steered by the dynamic-LDS request like the real kernels -- no LDS traffic, no shifts, no metadata.
// What a kernel loses against this stream on the SAME buffers in the SAME run is the kernel's; what the stream itself loses
// against the 8 TB/s peak is the memory system's (and the box's: DESIGN.md section 4).
#pragma once
#include "fl_widths.hpp"

namespace fl {

struct BareArgs {
    const char* in;
    const char* aux;
    char* out;
    uint64_t n_units;
    uint64_t tiles_per_xcd;
    unsigned in_unit, aux_unit, out_unit;    // bytes per unit, each a multiple of 16 and <= 8192 (aux_unit <= 1024)
    unsigned window_shift;
};

constexpr unsigned BARE_MAX_UNIT = 8192;

template <bool NT>
__global__ __launch_bounds__(WG) void k_bare_stream(BareArgs a)
{
    const uint64_t n_tiles = (a.n_units + (WG / 64) - 1) / (WG / 64);
    const uint64_t tile = xcd_tile(blockIdx.x, a.tiles_per_xcd, a.window_shift);
    if (tile >= n_tiles) return;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint64_t unit = tile * (WG / 64) + wave;
    if (unit >= a.n_units) return;
    const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.in) + unit * a.in_unit, 0, a.in_unit, 0x00020000);
    const __amdgpu_buffer_rsrc_t aux_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.aux) + unit * a.aux_unit, 0, a.aux_unit, 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(a.out + unit * a.out_unit, 0, a.out_unit, 0x00020000);
    u32x4 v[BARE_MAX_UNIT / 1024];
    static_for<BARE_MAX_UNIT / 1024>([&](auto G) {
        constexpr unsigned g = decltype(G)::value;
        v[g] = u32x4{0, 0, 0, 0};
        if (g * 1024u < a.in_unit) v[g] = __builtin_amdgcn_raw_buffer_load_b128(in_rs, lane * 16u + g * 1024u, 0, NT ? 2 : 0);   // wave-uniform
    });
    u32x4 acc = {lane, 0, 0, 0};
    if (a.aux_unit) acc = __builtin_amdgcn_raw_buffer_load_b128(aux_rs, lane * 16u, 0, 0);
    static_for<BARE_MAX_UNIT / 1024>([&](auto G) { acc ^= v[decltype(G)::value]; });
    static_for<BARE_MAX_UNIT / 1024>([&](auto G) {
        constexpr unsigned g = decltype(G)::value;
        if (g * 1024u < a.out_unit) __builtin_amdgcn_raw_buffer_store_b128(acc + g, out_rs, lane * 16u + g * 1024u, 0, STORE_AUX);
    });
    // a read-only stream keeps its loads alive through a store that never happens
    if (a.out_unit == 0 && acc.x == 0x12345678u && acc.y == 0x9abcdef0u && acc.z == 0x0fedcba9u) *reinterpret_cast<u32x4*>(a.out) = acc;
}

// window_log2_blocks: log2 of the tile-map window in units, WINDOW_WHOLE (31) = the whole-column map
inline hipError_t launch_bare_stream(BareArgs a, bool nt_loads, int waves, int window_log2_units, hipStream_t s)
{
    if (a.n_units == 0) return hipSuccess;
    if (a.in_unit > BARE_MAX_UNIT || a.out_unit > BARE_MAX_UNIT || a.aux_unit > 1024u || ((a.in_unit | a.out_unit | a.aux_unit) & 15u))
        return hipErrorInvalidValue;
    const uint64_t n_tiles = (a.n_units + (WG / 64) - 1) / (WG / 64);
    a.tiles_per_xcd = (n_tiles + 7) / 8;
    if (a.tiles_per_xcd * 8 > 0x7fffffffull) return hipErrorInvalidValue;
    if (window_log2_units >= WINDOW_WHOLE || window_log2_units <= 0) a.window_shift = 63u;
    else a.window_shift = (unsigned)(window_log2_units - 2 < 3 ? 3 : window_log2_units - 2);       // tiles of 4 units
    if (waves < 3) waves = 3;
    if (waves > 8) waves = 8;
    const unsigned lds = (CU_LDS_BYTES * (unsigned)WG / ((unsigned)waves * 256u)) & ~1023u;                                 // workgroups per CU = waves per SIMD
    if (nt_loads) FL_LAUNCH((k_bare_stream<true>), dim3((unsigned)(a.tiles_per_xcd * 8)), dim3(WG), lds, s, a);
    else FL_LAUNCH((k_bare_stream<false>), dim3((unsigned)(a.tiles_per_xcd * 8)), dim3(WG), lds, s, a);
    return hipGetLastError();
}

}  // namespace fl
