// fl_device.hpp -- device-side building blocks of the FastLanes codec for gfx950.
//
// Geometry (all element types).  A 1024-value block is a matrix of 16-byte
// CELLS, 8 cells (=128 B) per row:
//   * packed   : W rows  -- row w holds word w of every FL lane
//                (reference: `packed[LANES * w + lane]`, macros.rs:89/:158)
//   * unpacked : T rows  -- logical row r (the r-th value of every FL lane)
//                lives at cell-row  (r%8)*T + FL_ORDER[r/8]*sizeof(T)
//                (reference: index(row,lane), macros.rs:20-24)
// One GPU thread owns ONE CELL COLUMN c (0..7) of one block, i.e. 16/sizeof(T)
// adjacent FL lanes for ALL rows.  Consequences:
//   * every global access is a 16-byte dwordx4; 8 neighbouring threads cover a
//     full 128-byte row (one L2 line) of their block;
//   * row, word index, shift and mask are compile-time constants for every
//     thread -- exactly the property the reference gets from its unrolled macro;
//   * Delta's per-lane serial chain (delta.rs:56-61) is thread-local: no cross-lane
//     traffic (LDS is used wave-locally, only to shape global accesses: the row-store
//     staging of fl_kernels.hpp and the run exchange of the transposes below);
//   * u8/u16 lanes are processed SWAR inside 32-bit registers (masks are chosen
//     so no bit ever crosses an element boundary).
// A 64-lane wavefront therefore processes 8 blocks, a 256-thread workgroup 32.
//
// This header is also the user-extensible "functor" surface corresponding to
// the reference's exported pack!/unpack!/iterate! macros (macros.rs:11,34,100):
// iterate_rows<T>(f) walks the rows in order (iterate!), unpack_rows<T,W>(cells, f)
// calls f(row_constant, cell) in row order (unpack!), and pack_rows<T,W>(src_of_row,
// sink_of_word) is its inverse (pack!).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

namespace fl {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// Every kernel launch of the library.  hipGetLastError() reports the calling thread's most recent HIP error, whoever caused
// it: a host framework's benign failure before our launch (an event query that is not ready yet, a pointer-attribute lookup
// on plain host memory) would come back as OUR launch failing (seen in round 4: fl_fill_random returned FL_ERR_HIP as the
// first call after torch had created a stream).  So the slot is cleared first, and the launchers' closing hipGetLastError()
// reports this launch only.
#define FL_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

// lib.rs:22 FL_ORDER = [0,4,2,6,1,5,3,7], nibble-packed so it folds.
__host__ __device__ constexpr int fl_order(int o) { return (0x73516240u >> (4 * o)) & 7; }

template <int N, typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>)
{
    (f(std::integral_constant<int, Is>{}), ...);
}
// Compile-time unrolled loop: f(integral_constant<int,0>) ... f(integral_constant<int,N-1>).
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// ---------------------------------------------------------------------------
// Element traits (lib.rs:24-32: T = bits, LANES = 1024 / T)
// ---------------------------------------------------------------------------
template <typename T> struct Elem {
    static constexpr int BITS = sizeof(T) * 8;
    static constexpr int LANES = 1024 / BITS;
    static constexpr int PER_CELL = 16 / sizeof(T);   // FL lanes per 16-byte cell
    static constexpr int CELLS_PER_BLOCK = 8 * BITS;  // 1024*sizeof(T)/16
    // cell-row of logical row r inside an unpacked block (macros.rs:20-24 in cells)
    __host__ __device__ static constexpr int row_cell(int r)
    {
        return (r % 8) * BITS + fl_order(r / 8) * (int)sizeof(T);
    }
};

// A cell: 16 bytes = 4 dwords (u8/u16/u32) or 2 qwords (u64).
template <typename T> struct Cell {
    using word_t = std::conditional_t<sizeof(T) == 8, uint64_t, uint32_t>;
    static constexpr int NW = 16 / sizeof(word_t);
    static constexpr int TB = sizeof(T) * 8;
    word_t x[NW];

    // element mask replicated across the word (SWAR for u8/u16)
    __host__ __device__ static constexpr word_t rep(int bits)
    {
        const uint64_t m = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
        if (sizeof(T) == 1) return (word_t)((m & 0xffu) * 0x01010101u);
        if (sizeof(T) == 2) return (word_t)((m & 0xffffu) * 0x00010001u);
        return (word_t)m;
    }
    __device__ __forceinline__ static Cell zero()
    {
        Cell c;
        for (int i = 0; i < NW; ++i) c.x[i] = 0;
        return c;
    }
    __device__ __forceinline__ static Cell splat(T v)
    {
        Cell c;
        word_t w = (word_t)v;
        if (sizeof(T) == 1) w = (word_t)((uint32_t)(uint8_t)v * 0x01010101u);
        if (sizeof(T) == 2) w = (word_t)((uint32_t)(uint16_t)v * 0x00010001u);
        for (int i = 0; i < NW; ++i) c.x[i] = w;
        return c;
    }
    // (elem >> SH) & mask(BITS)   requires SH + BITS <= TB   (macros.rs:150,164)
    template <int SH, int BITS> __device__ __forceinline__ Cell extract() const
    {
        static_assert(SH + BITS <= TB, "field must lie inside the element");
        Cell c;
        for (int i = 0; i < NW; ++i) {
            if (BITS == 0) c.x[i] = 0;
            else if (SH + BITS == TB && sizeof(T) >= 4) c.x[i] = x[i] >> SH;
            else c.x[i] = (x[i] >> SH) & rep(BITS);
        }
        return c;
    }
    // (elem & mask(BITS)) << SH   requires SH + BITS <= TB   (macros.rs:79,160)
    template <int SH, int BITS> __device__ __forceinline__ Cell deposit() const
    {
        static_assert(SH + BITS <= TB, "field must lie inside the element");
        Cell c;
        for (int i = 0; i < NW; ++i) {
            if (BITS == 0) c.x[i] = 0;
            else if (SH + BITS == TB && sizeof(T) >= 4) c.x[i] = x[i] << SH;
            else c.x[i] = (x[i] & rep(BITS)) << SH;
        }
        return c;
    }
    __device__ __forceinline__ Cell operator|(const Cell& o) const
    {
        Cell c;
        for (int i = 0; i < NW; ++i) c.x[i] = x[i] | o.x[i];
        return c;
    }
    // element-wise wrapping add / sub
    __device__ __forceinline__ Cell add(const Cell& o) const
    {
        Cell c;
        for (int i = 0; i < NW; ++i) {
            if constexpr (sizeof(T) == 1) {
                const uint32_t H = 0x80808080u;
                c.x[i] = ((x[i] & ~H) + (o.x[i] & ~H)) ^ ((x[i] ^ o.x[i]) & H);
            } else if constexpr (sizeof(T) == 2) {
                u16x2 a = __builtin_bit_cast(u16x2, x[i]), b = __builtin_bit_cast(u16x2, o.x[i]);
                c.x[i] = __builtin_bit_cast(uint32_t, (u16x2)(a + b));   // v_pk_add_u16
            } else {
                c.x[i] = x[i] + o.x[i];
            }
        }
        return c;
    }
    __device__ __forceinline__ Cell sub(const Cell& o) const
    {
        Cell c;
        for (int i = 0; i < NW; ++i) {
            if constexpr (sizeof(T) == 1) {
                const uint32_t H = 0x80808080u;
                c.x[i] = ((x[i] | H) - (o.x[i] & ~H)) ^ ((x[i] ^ ~o.x[i]) & H);
            } else if constexpr (sizeof(T) == 2) {
                u16x2 a = __builtin_bit_cast(u16x2, x[i]), b = __builtin_bit_cast(u16x2, o.x[i]);
                c.x[i] = __builtin_bit_cast(uint32_t, (u16x2)(a - b));   // v_pk_sub_u16
            } else {
                c.x[i] = x[i] - o.x[i];
            }
        }
        return c;
    }
};

// 16-byte global accesses.  NT = non-temporal (streamed once, never re-read).
template <typename T, bool NT = false>
__device__ __forceinline__ Cell<T> load_cell(const u32x4* p)
{
    u32x4 v = NT ? __builtin_nontemporal_load(p) : *p;
    return __builtin_bit_cast(Cell<T>, v);
}
template <typename T, bool NT = false>
__device__ __forceinline__ void store_cell(u32x4* p, const Cell<T>& c)
{
    u32x4 v = __builtin_bit_cast(u32x4, c);
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// ---------------------------------------------------------------------------
// The FastLanes transpose (transpose.rs:9-36) seen from a cell-column thread.
// Along FL-lane l's row order the transposed positions index(r,l) map to T CONSECUTIVE
// original positions:
//     tau(index(r, l)) = lane_base(l) + r,   lane_base(l) = (l%16)*64 + FL_ORDER[l/16]*8
// (checked against transpose.rs:29-36 for every T, r, l in tests/test_oracle_properties.py).
// The thread that owns lanes n*c .. n*c+n-1 for all T rows therefore holds, per lane, one
// contiguous RUN of T elements (T*sizeof(T) bytes: 8 B for u8 .. 512 B for u64) of the
// original order: the 1024-element permutation needs no shuffle network, only a regroup of
// registers.  The one thing exchanged between the 8 threads of a block (through a wave-private
// LDS region, store_lane_runs_lines / load_lane_runs_lines below) is 16-byte pieces of those
// runs, purely so that the run-side global accesses are full 128-byte lines as well.
// ---------------------------------------------------------------------------
__host__ __device__ constexpr unsigned lane_base(unsigned l) { return (l % 16) * 64 + fl_order(l / 16) * 8; }

template <typename T> __device__ __forceinline__ uint64_t cell_get(const Cell<T>& c, int e)
{
    if constexpr (sizeof(T) == 8) return c.x[e];
    else if constexpr (sizeof(T) == 4) return c.x[e];
    else if constexpr (sizeof(T) == 2) return (c.x[e / 2] >> (16 * (e % 2))) & 0xffffu;
    else return (c.x[e / 4] >> (8 * (e % 4))) & 0xffu;
}
template <typename T> __device__ __forceinline__ void cell_or(Cell<T>& c, int e, uint64_t v)
{
    if constexpr (sizeof(T) == 8) c.x[e] = v;
    else if constexpr (sizeof(T) == 4) c.x[e] = (uint32_t)v;
    else if constexpr (sizeof(T) == 2) c.x[e / 2] |= (uint32_t)v << (16 * (e % 2));
    else c.x[e / 4] |= (uint32_t)v << (8 * (e % 4));
}

// store_lane_runs_lines: original-order block <- rows[r] (column c of a TRANSPOSED block), the
// inverse of load_lane_runs_lines, with every global store a FULL 128-byte line.  The 8 threads
// of a block exchange 16-byte pieces through a private 1152-byte LDS region: 8 line slots with a
// 144-byte stride (for u16, slots 4..7 sit a further 16 bytes on), which makes ds_write_b128 /
// ds_read_b128 conflict-free for u16/u32/u64.  One "phase" moves 8 output lines; a block has
// 8*sizeof(T) lines, hence sizeof(T) phases.  After the exchange thread c' stores piece c' of
// each line through `st` (a TileStore over the original-order block).  Only lanes of the same
// wavefront touch a region and the LDS operations of one wave execute in order, so a
// compiler-level fence is all the synchronisation needed (no s_barrier).
template <typename T> struct RunExchange {
    static constexpr int E = sizeof(T);
    static constexpr int PHASES = E;          // 8E lines per block, 8 per phase
    static constexpr int LS = 144;            // padded line stride in LDS
    static constexpr int BLOCK_BYTES = 8 * LS;
    static constexpr int WAVE_BYTES = 8 * BLOCK_BYTES;
    // LDS offset of line slot s inside a block's region
    __host__ __device__ static constexpr unsigned slot_base(int s) { return s * LS + (E == 2 ? 16 * (s / 4) : 0); }
    // byte offset, inside the ORIGINAL-order block, of the 128-byte line that slot s holds in phase p
    //   u64: phase = (lane e = p/4, quarter q = p%4) of the slot's thread s -> lane_base(2s+e)*8 + 128q
    //   u32: phase = lane e = p of thread s                                 -> lane_base(4s+p)*4
    //   u16: lines 8*(s/4) + 4p + s%4 (four 32-byte runs each);   u8: line s (sixteen 8-byte runs)
    __host__ __device__ static constexpr unsigned line_of(int p, int s)
    {
        return E == 8 ? (unsigned)((2 * s + p / 4) * 512 + 128 * (p % 4))
             : E == 4 ? (unsigned)(((4 * s + p) % 16) * 256 + ((4 * s + p) / 16) * 128)
             : E == 2 ? (unsigned)((8 * (s / 4) + 4 * p + s % 4) * 128)
                      : (unsigned)(s * 128);
    }
};

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <typename T, typename Store>
__device__ __forceinline__ void store_lane_runs_lines(char* lds_blk, unsigned c, const Cell<T>* rows, const Store& st)
{
    using X = RunExchange<T>;
    constexpr int E = X::E;
    static_for<X::PHASES>([&](auto P) {
        constexpr int p = decltype(P)::value;
        if constexpr (E >= 4) {
            // u32: phase = lane e, the lane's 128-byte run is one line.  u64: phase = (lane e, quarter q)
            constexpr int e = (E == 4) ? p : p / 4;
            constexpr int q = (E == 8) ? p % 4 : 0;
            constexpr int RPP = 16 / E;                       // rows per 16-byte piece
            static_for<8>([&](auto K) {
                constexpr int k = decltype(K)::value;
                constexpr int r0 = q * (128 / E) + k * RPP;
                u32x4 piece;
                if constexpr (E == 4) {
                    static_for<4>([&](auto D) { piece[decltype(D)::value] = (uint32_t)cell_get<T>(rows[r0 + decltype(D)::value], e); });
                } else {
                    const uint64_t a = cell_get<T>(rows[r0], e), b = cell_get<T>(rows[r0 + 1], e);
                    piece[0] = (uint32_t)a; piece[1] = (uint32_t)(a >> 32); piece[2] = (uint32_t)b; piece[3] = (uint32_t)(b >> 32);
                }
                *reinterpret_cast<u32x4*>(lds_blk + c * X::LS + 16 * k) = piece;
            });
        } else if constexpr (E == 2) {
            // 4 lanes per phase, 32-byte runs: line (l%16), quarter offset FL_ORDER[l/16]*16 bytes
            static_for<4>([&](auto EP) {
                constexpr int ep = decltype(EP)::value;
                constexpr int e = 4 * p + ep;
                static_for<2>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    u32x4 piece;
                    static_for<4>([&](auto D) {
                        constexpr int d = decltype(D)::value;
                        piece[d] = (uint32_t)cell_get<T>(rows[8 * j + 2 * d], e) | ((uint32_t)cell_get<T>(rows[8 * j + 2 * d + 1], e) << 16);
                    });
                    // slots 4..7 (written by the odd columns) sit 16 bytes further: conflict-free ds_write_b128
                    *reinterpret_cast<u32x4*>(lds_blk + (4 * (c & 1u) + ep) * X::LS + (c & 1u) * 16 + fl_order(c >> 1) * 16 + 16 * j) = piece;
                });
            });
        } else {
            // u8: one phase; lane e's 8-byte run sits in line e/2 at (e%2)*64 + FL_ORDER[c]*8
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            static_for<16>([&](auto EE) {
                constexpr int e = decltype(EE)::value;
                u32x2 piece;
                static_for<2>([&](auto D) {
                    constexpr int d = decltype(D)::value;
                    uint32_t w = 0;
                    static_for<4>([&](auto J) { w |= (uint32_t)cell_get<T>(rows[4 * d + decltype(J)::value], e) << (8 * decltype(J)::value); });
                    piece[d] = w;
                });
                *reinterpret_cast<u32x2*>(lds_blk + (e / 2) * X::LS + (e % 2) * 64 + fl_order(c) * 8) = piece;
            });
        }
        wave_lds_fence();
        static_for<8>([&](auto S) {
            constexpr int s = decltype(S)::value;
            const u32x4 piece = *reinterpret_cast<const u32x4*>(lds_blk + X::slot_base(s) + 16 * c);
            st.store(X::line_of(p, s) / 16, __builtin_bit_cast(Cell<T>, piece));   // st adds this thread's 16*c
        });
        wave_lds_fence();
    });
}

// load_lane_runs_lines: the mirror image -- rows[r] (column c of the TRANSPOSED block) <- the
// original-order block, reading it with FULL-line loads (thread c' fetches piece c' of each
// line; all 8*sizeof(T) loads are issued up front) and handing every thread its own lanes'
// runs through the same wave-private LDS exchange.
template <typename T>
__device__ __forceinline__ void load_lane_runs_lines(char* lds_blk, unsigned c, const u32x4* blk_cells, Cell<T>* rows)
{
    using X = RunExchange<T>;
    constexpr int E = X::E;
    constexpr int TB = Elem<T>::BITS;
    u32x4 lines[X::PHASES][8];
    static_for<X::PHASES>([&](auto P) {
        constexpr int p = decltype(P)::value;
        static_for<8>([&](auto S) {
            constexpr int s = decltype(S)::value;
            lines[p][s] = __builtin_nontemporal_load(blk_cells + X::line_of(p, s) / 16 + c);
        });
    });
    static_for<TB>([&](auto R) { rows[decltype(R)::value] = Cell<T>::zero(); });
    static_for<X::PHASES>([&](auto P) {
        constexpr int p = decltype(P)::value;
        static_for<8>([&](auto S) {
            *reinterpret_cast<u32x4*>(lds_blk + X::slot_base(decltype(S)::value) + 16 * c) = lines[p][decltype(S)::value];
        });
        wave_lds_fence();
        if constexpr (E >= 4) {
            constexpr int e = (E == 4) ? p : p / 4;
            constexpr int q = (E == 8) ? p % 4 : 0;
            constexpr int RPP = 16 / E;
            static_for<8>([&](auto K) {
                constexpr int k = decltype(K)::value;
                constexpr int r0 = q * (128 / E) + k * RPP;
                const u32x4 piece = *reinterpret_cast<const u32x4*>(lds_blk + c * X::LS + 16 * k);
                if constexpr (E == 4) {
                    static_for<4>([&](auto D) { cell_or<T>(rows[r0 + decltype(D)::value], e, piece[decltype(D)::value]); });
                } else {
                    cell_or<T>(rows[r0], e, (uint64_t)piece[0] | ((uint64_t)piece[1] << 32));
                    cell_or<T>(rows[r0 + 1], e, (uint64_t)piece[2] | ((uint64_t)piece[3] << 32));
                }
            });
        } else if constexpr (E == 2) {
            static_for<4>([&](auto EP) {
                constexpr int ep = decltype(EP)::value;
                constexpr int e = 4 * p + ep;
                static_for<2>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    const u32x4 piece = *reinterpret_cast<const u32x4*>(
                        lds_blk + (4 * (c & 1u) + ep) * X::LS + (c & 1u) * 16 + fl_order(c >> 1) * 16 + 16 * j);
                    static_for<4>([&](auto D) {
                        constexpr int d = decltype(D)::value;
                        cell_or<T>(rows[8 * j + 2 * d], e, piece[d] & 0xffffu);
                        cell_or<T>(rows[8 * j + 2 * d + 1], e, piece[d] >> 16);
                    });
                });
            });
        } else {
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            static_for<16>([&](auto EE) {
                constexpr int e = decltype(EE)::value;
                const u32x2 piece = *reinterpret_cast<const u32x2*>(lds_blk + (e / 2) * X::LS + (e % 2) * 64 + fl_order(c) * 8);
                static_for<8>([&](auto R) {
                    constexpr int r = decltype(R)::value;
                    cell_or<T>(rows[r], e, (piece[r / 4] >> (8 * (r % 4))) & 0xffu);
                });
            });
        }
        wave_lds_fence();
    });
}

// ---------------------------------------------------------------------------
// iterate_rows: the iterate! macro (macros.rs:11-32) on one cell column -- visits the block's T logical
// rows IN ORDER (the order a stateful body such as Delta's running value needs, delta.rs:24-45), handing the
// body the compile-time row and where that row's cell sits in an unpacked block:
//   f(R, C):  R = integral_constant<int, row>,  C = integral_constant<int, Elem<T>::row_cell(row)>
// i.e. the thread's column c of row `row` is cell `C::value + c` of the block (`idx = index(row, lane)` for the
// 16/sizeof(T) lanes of that cell at once).  unpack_rows / pack_rows below are the same walk with the bit
// (un)packing of unpack! / pack! spliced in.
// ---------------------------------------------------------------------------
template <typename T, typename F>
__device__ __forceinline__ void iterate_rows(F&& f)
{
    static_for<Elem<T>::BITS>([&](auto R) {
        f(R, std::integral_constant<int, Elem<T>::row_cell(decltype(R)::value)>{});
    });
}

// ---------------------------------------------------------------------------
// unpack_rows: the unpack! macro (macros.rs:100-174) on one cell column.
//   in[w]   : packed word-row w of this column (W cells, all in registers)
//   f(R, c) : called for R = integral_constant<int,row>, row = 0..T-1 in order
// ---------------------------------------------------------------------------
// One row of the unpack! expansion: the W-bit field of logical row ROW of this column.
template <typename T, int W, int ROW>
__device__ __forceinline__ Cell<T> unpack_row(const Cell<T>* in)
{
    constexpr int TB = Elem<T>::BITS;
    static_assert(W >= 0 && W <= TB, "BitPackWidth<W>: W <= T (bitpacking.rs:8-13)");
    static_assert(ROW >= 0 && ROW < TB, "row out of range");
    if constexpr (W == 0) {
        return Cell<T>::zero();                          // macros.rs:118-125
    } else if constexpr (W == TB) {
        return in[ROW];                                  // macros.rs:126-132
    } else {
        constexpr int curr = (ROW * W) / TB;             // macros.rs:144
        constexpr int next = ((ROW + 1) * W) / TB;       // macros.rs:145
        constexpr int shift = (ROW * W) % TB;            // macros.rs:147
        if constexpr (next > curr) {
            constexpr int rem = ((ROW + 1) * W) % TB;
            constexpr int cur = W - rem;
            Cell<T> v = in[curr].template extract<shift, cur>();       // macros.rs:152
            if constexpr (next < W && rem > 0)                         // macros.rs:156-161
                v = v | in[next].template deposit<cur, rem>();
            return v;
        } else {
            return in[curr].template extract<shift, W>();              // macros.rs:164
        }
    }
}

template <typename T, int W, typename F>
__device__ __forceinline__ void unpack_rows(const Cell<T>* in, F&& f)
{
    static_for<Elem<T>::BITS>([&](auto R) { f(R, unpack_row<T, W, decltype(R)::value>(in)); });
}

// Same rows, visited in ASCENDING ADDRESS order of the unpacked block instead of row order
// (cell-row j <-> logical row FL_ORDER[j % (T/8) * (64/T)] * 8 + j / (T/8); FL_ORDER is its own
// inverse, lib.rs:53-59).  Only for stateless bodies (plain / FoR stores): the reference
// visits rows in order "in case the kernel has side effects" (macros.rs:119), which a store
// to a distinct index has not.  Measured +1.5 % on u32 W=7 (profiles/abbench_r01h.txt).
template <typename T, int W, typename F>
__device__ __forceinline__ void unpack_rows_by_address(const Cell<T>* in, F&& f)
{
    constexpr int TB = Elem<T>::BITS;
    constexpr int PER_S = TB / 8;
    static_for<TB>([&](auto J) {
        constexpr int j = decltype(J)::value;
        constexpr int row = fl_order((j % PER_S) * (8 / PER_S)) * 8 + j / PER_S;
        static_assert(Elem<T>::row_cell(row) == 8 * j, "address-order row mapping (a row is 8 cells)");
        f(std::integral_constant<int, row>{}, unpack_row<T, W, row>(in));
    });
}

// ---------------------------------------------------------------------------
// pack_rows: the pack! macro (macros.rs:34-98) on one cell column.
//   src(R)     : returns the (already body-transformed) cell of logical row R
//   sink(Wd,c) : receives packed word-row Wd = integral_constant<int,w>
// W == 0 produces nothing (macros.rs:52-53); W == T copies unmasked (:54-59).
// ---------------------------------------------------------------------------
template <typename T, int W, typename S, typename K>
__device__ __forceinline__ void pack_rows(S&& src, K&& sink)
{
    constexpr int TB = Elem<T>::BITS;
    static_assert(W >= 0 && W <= TB, "BitPackWidth<W>: W <= T (bitpacking.rs:8-13)");
    if constexpr (W == 0) {
        return;
    } else if constexpr (W == TB) {
        static_for<TB>([&](auto R) { sink(R, src(R)); });
    } else {
        Cell<T> tmp = Cell<T>::zero();
        static_for<TB>([&](auto R) {
            constexpr int row = decltype(R)::value;
            constexpr int shift = (row * W) % TB;
            constexpr int curr = (row * W) / TB;
            constexpr int next = ((row + 1) * W) / TB;
            // bits of this value that stay in word `curr`; the rest is the carry.
            // (src & mask(W)) << shift drops bits past T (macros.rs:73,79); masking
            // to `keep` bits first gives the same word and keeps SWAR lanes apart.
            constexpr int keep = (shift + W <= TB) ? W : TB - shift;
            const Cell<T> s = src(R);
            if constexpr (row == 0) tmp = s.template deposit<0, keep>();
            else tmp = tmp | s.template deposit<shift, keep>();
            if constexpr (next > curr) {
                sink(std::integral_constant<int, curr>{}, tmp);        // macros.rs:89
                constexpr int rem = ((row + 1) * W) % TB;
                tmp = s.template extract<W - rem, rem>();              // macros.rs:92
            }
        });
    }
}

}  // namespace fl
