// fl_kernels.hpp -- batched gfx950 kernels over n_blocks contiguous 1024-value
// blocks, plus the runtime-width launch tables (the analogue of the reference's
// `match width { #(W => Self::unpack::<W>(..))* }`, bitpacking.rs:82-95,115-128).
//
// Thread mapping: global thread g -> block g/8, cell column g%8 (fl_device.hpp).
// Every kernel is a pure stream: each input byte is read once, each output byte
// written once, 16 B per lane per access; no s_barrier, no inter-workgroup
// communication (XCD placement is used for speed only).  LDS is used wave-locally
// and only to shape global stores (WaveRowStore below, RunExchange in fl_device.hpp).
#pragma once
#include "fl_device.hpp"
#include "fl_dispatch.hpp"
#include <atomic>

namespace fl {

#ifndef FL_WG
#define FL_WG 256                   // (-DFL_WG=128 / 64: A/B builds only -- make WGSIZE=128 -> libfastlanes_amd_wg128.so)
#endif
constexpr int WG = FL_WG;           // 4 wavefronts; 32 blocks ("one tile") per workgroup
constexpr int BLOCKS_PER_WG = WG / 8;

enum UnpackBody { BODY_STORE = 0, BODY_ADD_REF = 1, BODY_UNDELTA = 2,
                  // extension: undelta_pack fused with untranspose -- the decoded block is
                  // written in ORIGINAL order (= Transpose::untranspose(Delta::undelta_pack(..)), delta.rs:88-100)
                  BODY_UNDELTA_UNTRANSPOSE = 3 };
enum PackMode { PACK_PLAIN = 0, PACK_FOR = 1,
                // extension: pack(delta(transpose(v), base)) in one pass (the encode half of
                // delta.rs:88-95): reads the ORIGINAL-order block
                PACK_TRANSPOSE_DELTA = 2 };

// Kernel argument block shared by all streaming kernels.
struct StreamArgs {
    const u32x4* in;       // packed (unpack family) or unpacked (pack family)
    u32x4* out;
    const void* aux;       // references [n_blocks*aux_stride] or bases [n_blocks][LANES]
    uint64_t aux_stride;   // FoR: 0 = one scalar for all blocks, 1 = one per block
    uint64_t n_blocks;
    uint64_t tiles_per_xcd;   // ceil(ceil(n_blocks/32) / 8)
    unsigned window_shift;    // log2 of the tile-map window in tiles (>= 32: one window = the whole column) [| TILE_MAP_ROTATE]; see xcd_tile
};

// The XCD-aware tile map shared by every kernel: workgroup b -> tile.  The grid is 8 * tiles_per_xcd workgroups (a multiple
// of 8; padding workgroups get tiles past the end and leave).  It is walked in windows of 2^window_shift tiles (the last one
// shorter; every window a multiple of 8 tiles): workgroups [first, first + span) serve window [first, first + span), and
// inside it workgroup r -- which runs on XCD r % 8 (observed dispatch order; used for speed only, results never depend on
// it) -- takes tile first + (r % 8) * span / 8 + r / 8, so XCD x owns one contiguous eighth of the window.  One window =
// rounds 1-3's map (XCD x owns one contiguous eighth of the whole column).  Why windows, and which kernel gets one: fl_dispatch.hpp.
// TILE_MAP_ROTATE (a flag next to the window shift; round 4, profiles/abmixed_rotate_r04.txt): inside an XCD's run the k-th row of
// 32 tiles is rotated by k tiles.  An XCD has 32 CUs and its workgroups go to them round-robin, so WITHOUT the rotation CU c is
// handed tiles c, c + 32, c + 64, ... of the run: if the data has a period that divides 32 tiles (BASELINE config 5: width[b] =
// 1 + b mod 32 -- 8 tiles) every CU sits at ONE phase of the pattern for the whole launch, some decoding only wide blocks, others only
// narrow ones.  Rotated, every CU walks through all phases: the ramp 0.798 -> 0.812 (what seeded-random widths get), random widths
// -0.2 %.  Uniform-width kernels have nothing to decorrelate and lose 0.2-1 % to it, so only the mixed-width kernels set it.
constexpr unsigned TILE_MAP_ROTATE = 0x80u;
// (U = the integer type the map is computed in: uint64_t in general; a kernel whose slot count is known to fit 31 bits may ask for
// uint32_t -- the same code in scalar registers half as wide)
template <typename U> __device__ __forceinline__ U rotate_rows_of_32(U r, U run, bool rotate)
{
    const U row = r >> 5;
    return (rotate && ((row + 1) << 5) <= run) ? ((row << 5) | ((r + row) & 31u)) : r;      // a short last row stays as it is
}
template <typename U = uint64_t> __device__ __forceinline__ U xcd_tile(unsigned b, U tiles_per_xcd, unsigned window_shift_and_flags)
{
    // a window is a multiple of 8 tiles: anything below 2^3 (a launcher that skipped plan_grid / tile_window_shift) is taken as 2^3,
    // or several workgroups would map to one tile and others to none
    const unsigned window_shift = (window_shift_and_flags & 0x7fu) < 3u ? 3u : (window_shift_and_flags & 0x7fu);
    const bool rotate = (window_shift_and_flags & TILE_MAP_ROTATE) != 0;
    if (window_shift >= 32) return (U)(b & 7u) * tiles_per_xcd + rotate_rows_of_32<U>((U)(b >> 3), tiles_per_xcd, rotate);
    const unsigned first = (b >> window_shift) << window_shift;
    const unsigned r = b - first;
    const U left = tiles_per_xcd * 8 - first, full = (U)1 << window_shift;
    const U span = left < full ? left : full;
    return first + (U)(r & 7u) * (span >> 3) + rotate_rows_of_32<U>((U)(r >> 3), span >> 3, rotate);
}

// A/B tools (fl_internal_set_kernel_policy bits 25-29): log2 of the window in blocks for EVERY kernel; 0 = each kernel's default
inline std::atomic<int>& window_override()
{
    static std::atomic<int> v{0};
    return v;
}
// the same for the calling thread's launches only (the memory-class probe of fl_capi.hip); wins over the process-wide one
inline int& window_override_this_thread()
{
    static thread_local int v = 0;
    return v;
}
// set by every device-tier entry point of the C ABI (fl_capi.hip: FL_DEVICE_TIER) for the launches it makes: the call's buffers lie
// inside one live FL_LAYOUT_INTERLEAVED pair.  Such a pair has its input inside one class of memory and its output rotating through
// classes by the whole-column map's write positions; the table's windows are for plain allocations (they keep the eight XCDs' reads
// inside one class) and lose 1-4 % here on every row that has one (profiles/r06_window_matrix_constructed.txt).
inline bool& constructed_pair_this_thread()
{
    static thread_local bool v = false;
    return v;
}
// window_shift of a launch: the (op, type)'s window from the generated table (or the override) in blocks -> tiles of `tile_blocks` blocks
inline unsigned tile_window_shift(WindowOp op, unsigned type_bits, unsigned tile_blocks)
{
    const int mine = window_override_this_thread();
    const int ov = mine ? mine : window_override().load(std::memory_order_relaxed);
    const int lg = ov ? ov : constructed_pair_this_thread() ? (int)WINDOW_WHOLE : window_log2_blocks(op, type_bits);
    if (lg >= WINDOW_WHOLE) return 63u;
    int tl = 0;
    while ((2u << tl) <= tile_blocks) ++tl;                  // floor(log2(tile_blocks))
    const int sh = lg - tl;
    return (unsigned)(sh < 3 ? 3 : sh);                      // a window is a multiple of 8 tiles
}

// ---------------------------------------------------------------------------
// Streaming policy, fixed by measurement on MI355X (profiles/abbench_r01*.txt):
//  * XCD-contiguous tiles: workgroup b runs on XCD b%8 (observed dispatch order,
//    used for speed only), so XCD x is handed the contiguous eighth
//    [x*tiles_per_xcd, (x+1)*tiles_per_xcd) of the column instead of every 8th
//    tile: +1..11 % (each XCD's L2/fabric path sees one dense stream).
//  * Stores are write-through, non-temporal (`sc1 nt`, buffer-store aux 18):
//    output is never re-read, so lines should not linger dirty in L2: +6..7 %.
//  * Few waves in flight: 2-3 waves/SIMD (unpack) or 1 (pack, which already has
//    T x 16 B of loads in flight per lane) beat full occupancy by 2..4 % --
//    fewer concurrent DRAM streams.  Enforced with amdgpu_waves_per_eu.
//  * Loads are non-temporal when the read side is large (pack; unpack W >= T/2).
//  * Whole unpacked rows leave in ascending address order through a wave-private
//    LDS staging buffer, 1 KiB contiguous per store instruction (WaveRowStore):
//    +1.7..3.4 % and less box-to-box spread.
// ---------------------------------------------------------------------------
// nt (2) | sc1 (16).  Round 3 re-tried 2, 3, 16, 17 and 19 on the wave-per-block kernels (separate processes, so +-3 % of
// placement noise): none stands out against 18; plain sc1 (16) is 2 % behind.
#ifndef FL_STORE_AUX
#define FL_STORE_AUX 18             // (-DFL_STORE_AUX=n: A/B builds only -- make STOREAUX=n -> libfastlanes_amd_st<n>.so)
#endif
constexpr int STORE_AUX = FL_STORE_AUX;

// (u8 / u16 at 4, 5 and 8 waves per SIMD measured equal or worse on the same buffers, round 3; only u8 W=1 gained.)
template <typename T, int W> struct UnpackPolicy {
    static constexpr int MAXW = 2;          // bodies that keep all T rows in registers, and the mixed path
    static constexpr int MAXW_ROWS = 3;     // stateless row-at-a-time bodies (store / FoR)
    static constexpr bool NT_LOAD = (2 * W >= Elem<T>::BITS);
};
template <typename T> struct PackPolicy {
    static constexpr int MAXW = sizeof(T) >= 4 ? 1 : 2;
    static constexpr bool NT_LOAD = true;
};

// Tile (workgroup) index under the XCD-contiguous map; returns false for padding workgroups.
__device__ __forceinline__ bool tile_of_workgroup(const StreamArgs& a, uint64_t& tile)
{
    const uint64_t n_tiles = (a.n_blocks + BLOCKS_PER_WG - 1) / BLOCKS_PER_WG;
    tile = xcd_tile(blockIdx.x, a.tiles_per_xcd, a.window_shift);
    return tile < n_tiles;
}

// Write-through streaming store window over one tile's output (wave-uniform descriptor).
template <int BLOCK_BYTES> struct TileStore {
    __amdgpu_buffer_rsrc_t rs;
    unsigned vo;
    __device__ __forceinline__ TileStore(u32x4* out, uint64_t tile, uint64_t n_blocks, unsigned tid)
    {
        const uint64_t rem = n_blocks - tile * BLOCKS_PER_WG;
        const unsigned nrec = (unsigned)(rem < BLOCKS_PER_WG ? rem : BLOCKS_PER_WG) * BLOCK_BYTES;
        char* base = reinterpret_cast<char*>(out) + tile * (uint64_t)(BLOCKS_PER_WG * BLOCK_BYTES);
        rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, nrec, 0x00020000);
        vo = (tid >> 3) * BLOCK_BYTES + (tid & 7u) * 16;
    }
    template <typename T> __device__ __forceinline__ void store(unsigned cell_in_block, const Cell<T>& v) const
    {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, vo + 16 * cell_in_block, 0, STORE_AUX);
    }
};

// Wave-contiguous streaming stores for whole unpacked rows.  The cell-column mapping gives a
// store instruction 8 x 128 B (one line of each of the wave's 8 blocks, 4 KiB apart).  Staging
// 8 address-rows of every block (8 KiB per wave) through LDS turns that into 8 instructions
// of 1 KiB CONTIGUOUS bytes each (one block per instruction): +1.7..3.4 % on u32 W=7 and a
// smaller spread between boxes (profiles/abbench_r01k.txt).  ds_write_b128 / ds_read_b128 are
// both conflict-free (8 lanes x 16 B = one 128-byte row; 64 lanes x 16 B = 1 KiB linear); the
// exchange is wave-local, so it needs no s_barrier.
template <typename T> struct WaveRowStore {
    static constexpr unsigned BLOCK_BYTES = Elem<T>::CELLS_PER_BLOCK * 16;
    static constexpr int GROUPS = Elem<T>::BITS / 8;       // groups of 8 address-rows (1 KiB per block)
    static constexpr int WAVE_LDS = 8 * 1024;
    __amdgpu_buffer_rsrc_t rs;
    char* lds;
    unsigned lane;
    __device__ __forceinline__ WaveRowStore(u32x4* out, uint64_t first_blk, uint64_t n_blocks, char* wave_lds, unsigned lane_)
    {
        const uint64_t rem = n_blocks - first_blk;
        rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(out) + first_blk * BLOCK_BYTES, 0,
                                               (unsigned)(rem < 8 ? rem : 8) * BLOCK_BYTES, 0x00020000);
        lds = wave_lds;
        lane = lane_;
    }
    // address-row j of this thread's block (j = 8*group + i): cell of column c
    template <int I> __device__ __forceinline__ void put(const Cell<T>& v) const
    {
        *reinterpret_cast<u32x4*>(lds + (lane >> 3) * 1024 + I * 128 + (lane & 7u) * 16) = __builtin_bit_cast(u32x4, v);
    }
    template <int GROUP> __device__ __forceinline__ void flush() const
    {
        wave_lds_fence();
        static_for<8>([&](auto B) {
            constexpr int b = decltype(B)::value;
            const u32x4 v = *reinterpret_cast<const u32x4*>(lds + b * 1024 + lane * 16);
            // blocks past the end of the column fall outside the descriptor and are dropped
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, b * BLOCK_BYTES + GROUP * 1024 + lane * 16, 0, STORE_AUX);
        });
        wave_lds_fence();
    }
    // logical row stored at address-row j
    __host__ __device__ static constexpr int row_at(int j)
    {
        constexpr int PER_S = Elem<T>::BITS / 8;
        return fl_order((j % PER_S) * (8 / PER_S)) * 8 + j / PER_S;
    }
};

// unpack / unfor_pack / undelta_pack  (bitpacking.rs:98-107, ffor.rs:38-50,
// delta.rs:47-63): packed W cell-rows -> T cell-rows.
template <typename T, int W, int BODY>
__global__ __launch_bounds__(WG)
__attribute__((amdgpu_waves_per_eu(1, (BODY == BODY_STORE || BODY == BODY_ADD_REF) ? UnpackPolicy<T, W>::MAXW_ROWS
                                                                                    : UnpackPolicy<T, W>::MAXW)))
void k_unpack(StreamArgs a)
{
    constexpr bool NTL = UnpackPolicy<T, W>::NT_LOAD;
    constexpr int TB = Elem<T>::BITS;
    uint64_t tile;
    if (!tile_of_workgroup(a, tile)) return;
    const unsigned tid = threadIdx.x;
    const unsigned c = tid & 7u;

    if constexpr (BODY == BODY_UNDELTA_UNTRANSPOSE) {
        const uint64_t blk = tile * BLOCKS_PER_WG + (tid >> 3);
        if (blk >= a.n_blocks) return;
        Cell<T> in[W ? W : 1];
        const u32x4* pk = a.in + blk * (uint64_t)(8 * W) + c;
        static_for<W>([&](auto Wd) { in[decltype(Wd)::value] = load_cell<T, NTL>(pk + 8 * decltype(Wd)::value); });
        const TileStore<Elem<T>::CELLS_PER_BLOCK * 16> st(a.out, tile, a.n_blocks, tid);
        const u32x4* bases = static_cast<const u32x4*>(a.aux);
        Cell<T> prev = load_cell<T, false>(bases + blk * 8 + c);
        Cell<T> rows[TB];
        unpack_rows<T, W>(in, [&](auto R, const Cell<T>& v) {
            prev = v.add(prev);
            rows[decltype(R)::value] = prev;
        });
        __shared__ __attribute__((aligned(16))) char lds[(WG / 64) * RunExchange<T>::WAVE_BYTES];
        store_lane_runs_lines<T>(lds + (tid >> 3) * RunExchange<T>::BLOCK_BYTES, c, rows, st);
    } else {
        // rows leave through the wave-contiguous LDS staging: every lane takes part in the stores of
        // all 8 blocks of its wavefront, so only whole wavefronts past the end may leave early
        using WS = WaveRowStore<T>;
        __shared__ __attribute__((aligned(16))) char lds[(WG / 64) * WS::WAVE_LDS];
        // readfirstlane: the wave index must be provably wave-uniform, or hipcc wraps every buffer store
        // of the wave's descriptor in a waterfall loop (cdna_hip_programming.md T20)
        const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
        const uint64_t first_blk = tile * BLOCKS_PER_WG + wave * 8;
        if (first_blk >= a.n_blocks) return;
        const uint64_t blk = first_blk + (lane >> 3);
        const bool valid = blk < a.n_blocks;
        Cell<T> in[W ? W : 1];
        static_for<(W ? W : 1)>([&](auto Wd) { in[decltype(Wd)::value] = Cell<T>::zero(); });
        if (valid) {
            const u32x4* pk = a.in + blk * (uint64_t)(8 * W) + c;
            static_for<W>([&](auto Wd) { in[decltype(Wd)::value] = load_cell<T, NTL>(pk + 8 * decltype(Wd)::value); });
        }
        const WS ws(a.out, first_blk, a.n_blocks, lds + wave * WS::WAVE_LDS, lane);
        if constexpr (BODY == BODY_STORE || BODY == BODY_ADD_REF) {
            Cell<T> ref = Cell<T>::zero();
            if constexpr (BODY == BODY_ADD_REF) {
                if (valid) ref = Cell<T>::splat(static_cast<const T*>(a.aux)[blk * a.aux_stride]);
            }
            // stateless bodies: rows are produced directly in ascending address order (the reference
            // visits them in row order only "in case the kernel has side effects", macros.rs:119)
            static_for<WS::GROUPS>([&](auto G) {
                constexpr int grp = decltype(G)::value;
                static_for<8>([&](auto I) {
                    constexpr int row = WS::row_at(8 * grp + decltype(I)::value);
                    const Cell<T> v = unpack_row<T, W, row>(in);
                    if constexpr (BODY == BODY_ADD_REF) ws.template put<decltype(I)::value>(v.add(ref));   // ffor.rs:46-48
                    else ws.template put<decltype(I)::value>(v);                                          // bitpacking.rs:103-105
                });
                ws.template flush<grp>();
            });
        } else {
            // undelta_pack: the per-lane running sum needs row order (delta.rs:56-61); all T rows are
            // kept in registers and leave in address order afterwards
            Cell<T> prev = Cell<T>::zero();
            if (valid) prev = load_cell<T, false>(static_cast<const u32x4*>(a.aux) + blk * 8 + c);   // delta.rs:56
            Cell<T> rows[TB];
            unpack_rows<T, W>(in, [&](auto R, const Cell<T>& v) {
                prev = v.add(prev);                                                                  // delta.rs:58-60
                rows[decltype(R)::value] = prev;
            });
            static_for<WS::GROUPS>([&](auto G) {
                constexpr int grp = decltype(G)::value;
                static_for<8>([&](auto I) { ws.template put<decltype(I)::value>(rows[WS::row_at(8 * grp + decltype(I)::value)]); });
                ws.template flush<grp>();
            });
        }
    }
}

// pack / for_pack  (bitpacking.rs:65-74, ffor.rs:24-36): T cell-rows -> W cell-rows.
template <typename T, int W, int MODE>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(1, PackPolicy<T>::MAXW)))
void k_pack(StreamArgs a)
{
    constexpr int TB = Elem<T>::BITS;
    constexpr bool NTL = PackPolicy<T>::NT_LOAD;
    uint64_t tile;
    if (!tile_of_workgroup(a, tile)) return;
    const unsigned tid = threadIdx.x;
    const uint64_t blk = tile * BLOCKS_PER_WG + (tid >> 3);
    const unsigned c = tid & 7u;
    if (blk >= a.n_blocks) return;
    if constexpr (W == 0) return;                                               // macros.rs:52-53

    const u32x4* un = a.in + blk * (uint64_t)Elem<T>::CELLS_PER_BLOCK + c;
    Cell<T> ref = Cell<T>::zero();
    if constexpr (MODE == PACK_FOR) ref = Cell<T>::splat(static_cast<const T*>(a.aux)[blk * a.aux_stride]);

    // Issue all T row loads up front (they are independent), then combine.
    Cell<T> rows[TB];
    if constexpr (MODE == PACK_TRANSPOSE_DELTA) {
        __shared__ __attribute__((aligned(16))) char lds[(WG / 64) * RunExchange<T>::WAVE_BYTES];
        load_lane_runs_lines<T>(lds + (tid >> 3) * RunExchange<T>::BLOCK_BYTES, c,
                                a.in + blk * (uint64_t)Elem<T>::CELLS_PER_BLOCK, rows);                   // transpose.rs:12-14
        Cell<T> prev = load_cell<T, false>(static_cast<const u32x4*>(a.aux) + blk * 8 + c);
        static_for<TB>([&](auto R) {                                            // delta.rs:26-31
            const Cell<T> next = rows[decltype(R)::value];
            rows[decltype(R)::value] = next.sub(prev);
            prev = next;
        });
    } else {
        static_for<TB>([&](auto J) {   // issued in ascending address order
            rows[WaveRowStore<T>::row_at(decltype(J)::value)] = load_cell<T, NTL>(un + 8 * decltype(J)::value);
        });
    }
    const TileStore<(W ? W : 1) * 128> st(a.out, tile, a.n_blocks, tid);
    pack_rows<T, W>(
        [&](auto R) {
            if constexpr (MODE == PACK_FOR) return rows[decltype(R)::value].sub(ref);   // ffor.rs:32-34
            else return rows[decltype(R)::value];                               // bitpacking.rs:70-72
        },
        [&](auto Wd, const Cell<T>& v) { st.store(8 * decltype(Wd)::value, v); });
}

// delta / undelta  (delta.rs:24-45): T cell-rows -> T cell-rows, per-lane chain.
template <typename T, bool INVERSE>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(1, PackPolicy<T>::MAXW)))
void k_delta(StreamArgs a)
{
    constexpr int TB = Elem<T>::BITS;
    using WS = WaveRowStore<T>;
    __shared__ __attribute__((aligned(16))) char lds[(WG / 64) * WS::WAVE_LDS];
    uint64_t tile;
    if (!tile_of_workgroup(a, tile)) return;
    const unsigned tid = threadIdx.x;
    // readfirstlane: keeps the wave's store descriptor in SGPRs (no waterfall loop around each store)
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u, c = tid & 7u;
    const uint64_t first_blk = tile * BLOCKS_PER_WG + wave * 8;
    if (first_blk >= a.n_blocks) return;                       // whole wavefront past the end
    const uint64_t blk = first_blk + (lane >> 3);
    const bool valid = blk < a.n_blocks;
    Cell<T> rows[TB];
    Cell<T> prev = Cell<T>::zero();
    static_for<TB>([&](auto R) { rows[decltype(R)::value] = Cell<T>::zero(); });
    if (valid) {
        const u32x4* src = a.in + blk * (uint64_t)Elem<T>::CELLS_PER_BLOCK + c;
        prev = load_cell<T, false>(static_cast<const u32x4*>(a.aux) + blk * 8 + c);
        static_for<TB>([&](auto J) {   // issued in ascending address order
            rows[WaveRowStore<T>::row_at(decltype(J)::value)] = load_cell<T, true>(src + 8 * decltype(J)::value);
        });
    }
    iterate_rows<T>([&](auto R, auto) {                                         // iterate!(T, lane, |idx| ..), delta.rs:26,38
        constexpr int row = decltype(R)::value;
        if constexpr (INVERSE) {
            prev = rows[row].add(prev);                                         // delta.rs:40-42
            rows[row] = prev;
        } else {
            const Cell<T> next = rows[row];
            rows[row] = next.sub(prev);                                         // delta.rs:28-30
            prev = next;
        }
    });
    const WS ws(a.out, first_blk, a.n_blocks, lds + wave * WS::WAVE_LDS, lane);
    static_for<WS::GROUPS>([&](auto G) {
        constexpr int grp = decltype(G)::value;
        static_for<8>([&](auto I) { ws.template put<decltype(I)::value>(rows[WS::row_at(8 * grp + decltype(I)::value)]); });
        ws.template flush<grp>();
    });
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
typedef hipError_t (*stream_launch_t)(const StreamArgs&, hipStream_t);

// grid = 8 XCD slots x tiles_per_xcd (padding workgroups exit immediately)
// (user kernels on the functor API: plan_grid(a) = the whole-column map; plan_grid MUST fill the launch's tile-map fields)
inline unsigned plan_grid(StreamArgs& a, WindowOp op = WIN_TRANSPOSE, unsigned type_bits = 0)
{
    const uint64_t n_tiles = (a.n_blocks + BLOCKS_PER_WG - 1) / BLOCKS_PER_WG;
    a.tiles_per_xcd = (n_tiles + 7) / 8;
    a.window_shift = type_bits ? tile_window_shift(op, type_bits, BLOCKS_PER_WG) : 63u;
    return (unsigned)(a.tiles_per_xcd * 8);
}

template <typename T, int W, int BODY>
hipError_t launch_unpack(const StreamArgs& a0, hipStream_t s)
{
    if (a0.n_blocks == 0) return hipSuccess;
    StreamArgs a = a0;
    const unsigned grid = plan_grid(a, BODY == BODY_UNDELTA ? WIN_UNDELTA_PACK : BODY == BODY_UNDELTA_UNTRANSPOSE ? WIN_UNDELTA_PACK_UNTRANSPOSE : WIN_UNPACK, Elem<T>::BITS);
    FL_LAUNCH((k_unpack<T, W, BODY>), dim3(grid), dim3(WG), 0, s, a);
    return hipGetLastError();
}
template <typename T, int W, int MODE>
hipError_t launch_pack(const StreamArgs& a0, hipStream_t s)
{
    if (a0.n_blocks == 0 || W == 0) return hipSuccess;
    StreamArgs a = a0;
    const unsigned grid = plan_grid(a, MODE == PACK_TRANSPOSE_DELTA ? WIN_TRANSPOSE_DELTA_PACK : WIN_PACK, Elem<T>::BITS);
    FL_LAUNCH((k_pack<T, W, MODE>), dim3(grid), dim3(WG), 0, s, a);
    return hipGetLastError();
}
template <typename T, bool INVERSE>
hipError_t launch_delta(const StreamArgs& a0, hipStream_t s)
{
    if (a0.n_blocks == 0) return hipSuccess;
    StreamArgs a = a0;
    const unsigned grid = plan_grid(a, INVERSE ? WIN_UNDELTA : WIN_DELTA, Elem<T>::BITS);
    FL_LAUNCH((k_delta<T, INVERSE>), dim3(grid), dim3(WG), 0, s, a);
    return hipGetLastError();
}

// Runtime width -> instance table, index 0..T inclusive.
template <typename T> struct WidthTable { stream_launch_t fn[Elem<T>::BITS + 1]; };

// Only the (T, W) instances the dispatch table actually sends calls to are built (fl_dispatch.hpp: cell_column_built);
// the other entries are nullptr and the C ABI serves those widths with the runtime-width wave-per-block kernels.
constexpr WaveOp wave_op_of_body(int body)
{
    return body == BODY_UNDELTA ? WAVE_UNDELTA_PACK : body == BODY_UNDELTA_UNTRANSPOSE ? WAVE_UNDELTA_PACK_UNTRANSPOSE
         : body == BODY_ADD_REF ? WAVE_UNFOR_PACK : WAVE_UNPACK;
}
constexpr WaveOp wave_op_of_mode(int mode) { return mode == PACK_TRANSPOSE_DELTA ? WAVE_TRANSPOSE_DELTA_PACK : mode == PACK_FOR ? WAVE_FOR_PACK : WAVE_PACK; }

template <typename T, int W, int BODY> constexpr stream_launch_t unpack_entry()
{
    if constexpr (cell_column_built(Elem<T>::BITS, W, wave_op_of_body(BODY))) return &launch_unpack<T, W, BODY>;
    else return nullptr;
}
template <typename T, int W, int MODE> constexpr stream_launch_t pack_entry()
{
    if constexpr (cell_column_built(Elem<T>::BITS, W, wave_op_of_mode(MODE))) return &launch_pack<T, W, MODE>;
    else return nullptr;
}
template <typename T, int BODY, int... Ws>
constexpr WidthTable<T> make_unpack_table(std::integer_sequence<int, Ws...>)
{
    return WidthTable<T>{{unpack_entry<T, Ws, BODY>()...}};
}
template <typename T, int MODE, int... Ws>
constexpr WidthTable<T> make_pack_table(std::integer_sequence<int, Ws...>)
{
    return WidthTable<T>{{pack_entry<T, Ws, MODE>()...}};
}

// Specialised once per (element type, family) in fl_inst.hip
template <typename T, int BODY> const WidthTable<T>& unpack_table_impl();
template <typename T, int MODE> const WidthTable<T>& pack_table_impl();
template <typename T> stream_launch_t delta_launcher(bool inverse);

}  // namespace fl
