/*
 * fastlanes_amd.h -- C ABI of the MI355X-native FastLanes codec (libfastlanes_amd.so).
 *
 * The drop-in boundary for spiraldb/fastlanes' BitPacking / FoR / Delta /
 * Transpose trait methods (reference: /root/reference/src).  The Rust traits
 * are static methods monomorphised per element type and const width W; an FFI
 * cannot carry const generics, so every entry point takes the runtime `width`
 * exactly like the reference's own `unchecked_*` methods do
 * (bitpacking.rs:30,44,58) and dispatches to a per-(T,W) gfx950 kernel.
 *
 * Naming:  fl_<ty>_<method>[_host],  <ty> in {u8,u16,u32,u64},  <method> the
 * reference's method name.  Each function cites the reference interface it
 * replaces.
 *
 * Two tiers, same kernels:
 *   DEVICE tier  fl_<ty>_<method>(..., n_blocks, stream)
 *       All data pointers are DEVICE pointers (HBM).  The op is applied to
 *       n_blocks contiguous 1024-value blocks: block b reads
 *       in + b*in_block_elems and writes out + b*out_block_elems, where an
 *       unpacked block is 1024 elements and a packed block is 1024*W/T elements
 *       (= 128*W bytes; bitpacking.rs:77).  This is the reference's caller loop
 *       (benches/bitpacking.rs:80-97) moved on-device.  Asynchronous on `stream`
 *       (a hipStream_t, passed as void*; NULL = default stream); never
 *       synchronises, never allocates, retains no pointer.
 *   HOST tier    fl_<ty>_<method>_host(..., n_blocks)
 *       Same semantics with HOST pointers (the trait methods' `&[T]` slices,
 *       e.g. bitpacking.rs:19,33): runs the same kernels on the calling
 *       thread's current device and synchronises before returning.  n_blocks = 1
 *       is the exact shape of one trait-method call.  Small calls are zero-copy
 *       (the kernel reads/writes a pinned host buffer over PCIe; the calling
 *       thread polls a completion word in pinned memory for the ~10 us the call
 *       takes, whatever the device's scheduling flags say); large calls
 *       stage through a cached device buffer; after a thread's first call
 *       nothing is allocated (fl_host_release).  There is no CPU code path:
 *       without a GPU these return FL_ERR_HIP.
 *
 * Preconditions.  Device pointers 16-byte aligned (every block is a multiple
 * of 128 bytes, so block starts stay aligned; 128-byte alignment of the
 * column base is recommended for full-line accesses).  in/out must not
 * overlap.  width <= T.  Outputs are fully overwritten (pack with width 0
 * writes nothing, macros.rs:52-53; unpack with width 0 writes 1024 zeros per
 * block, macros.rs:118-125).
 *
 * Errors.  The reference has no Result type: `unchecked_*` panics via
 * unreachable!() on width > T (bitpacking.rs:93,126,197) and unpack_single
 * asserts index < 1024 (bitpacking.rs:152).  Here every function returns an
 * fl_status; a binding maps nonzero to panic! to match.
 * Nothing throws or aborts across the ABI.  Every launch first CLEARS the calling
 * thread's pending HIP error (hipGetLastError) so that FL_ERR_HIP always means this
 * call's launch failed, never an earlier benign failure of the application's own
 * HIP calls: an application that relies on hipGetLastError across a call into this
 * library must read it before the call.
 *
 * Threading and device selection.  Re-entrant and thread-safe; no global state
 * besides HIP's own, and no entry point ever calls hipSetDevice.  The device is
 * selected the way HIP selects it: kernels run on the device that is CURRENT for
 * the calling thread, so `stream` (when not NULL) must be a stream of that device
 * and every device pointer must be memory that device can use (its own HBM,
 * managed memory, or pinned host memory).  One host thread per device, each after
 * its own hipSetDevice, is the intended multi-GPU shape (examples/multi_gpu_decode.c).
 * A mismatch -- device 0 current, buffers or stream of device 1 -- is undefined
 * behaviour by default, exactly as for a raw kernel launch; with the environment
 * variable FL_CHECK_DEVICE=1 (read once, at the first device-tier call) every
 * device-tier entry point verifies it with hipPointerGetAttributes /
 * hipStreamGetDevice before launching and returns FL_ERR_DEVICE instead: a debug
 * aid that costs a few microseconds per call.
 */
#ifndef FASTLANES_AMD_H
#define FASTLANES_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum fl_status {
    FL_OK = 0,
    FL_ERR_WIDTH = 1,   /* width > T              (bitpacking.rs:93 unreachable!) */
    FL_ERR_INDEX = 2,   /* index >= 1024*n_blocks (bitpacking.rs:152 assert!)      */
    FL_ERR_NULL = 3,    /* required pointer is NULL                                */
    FL_ERR_ALIGN = 4,   /* device pointer not 16-byte aligned (fl_fill_random: 8)  */
    FL_ERR_HIP = 5,     /* HIP runtime error; see fl_last_hip_error()              */
    FL_ERR_BOUNDS = 6,  /* a block's bytes lie outside the packed column           */
    FL_ERR_DEVICE = 7   /* FL_CHECK_DEVICE=1 only: a device-tier pointer is not memory the
                           calling thread's current device can use, or `stream` belongs to
                           another device (see "Threading")                        */
} fl_status;

/* Device-side error bits.  The device tier never synchronises, so what the reference reports by panicking in the middle
 * of a caller loop (bitpacking.rs:93,126,152,197 and the debug_asserts on lengths :78-80,111-113,185-186) is reported
 * through *err_flag (a device uint32 the caller zeroes; may be NULL): the offending block / index is SKIPPED (a lookup
 * writes 0) and its bit is ORed in. */
enum {
    FL_DEVERR_WIDTH = 1,   /* widths[b] > T                                              -> FL_ERR_WIDTH  */
    FL_DEVERR_INDEX = 2,   /* lookup index >= 1024*n_blocks                              -> FL_ERR_INDEX  */
    FL_DEVERR_ALIGN = 4,   /* offsets[b] not a multiple of 16 (lookups: of sizeof(T))    -> FL_ERR_ALIGN  */
    FL_DEVERR_BOUNDS = 8   /* offsets[b] + 128*widths[b] > packed_bytes                  -> FL_ERR_BOUNDS */
};

/* predicates of fl_<ty>_unpack_compare (unsigned comparison with a constant) */
typedef enum fl_cmp { FL_CMP_EQ = 0, FL_CMP_NE = 1, FL_CMP_LT = 2, FL_CMP_LE = 3, FL_CMP_GT = 4, FL_CMP_GE = 5 } fl_cmp;

/* Library / build identification and diagnostics. */
const char *fl_version(void);
const char *fl_status_string(int status);
/* hipError_t of the most recent FL_ERR_HIP on the calling thread (0 if none). */
int fl_last_hip_error(void);
/* Elements in one packed block: 1024*width/T (bitpacking.rs:77); 0 if width > T. */
size_t fl_packed_len(unsigned type_bits, unsigned width);
/* The host tier keeps one cached context per calling thread (a private stream, a pinned staging
 * buffer, a device scratch buffer; see "HOST tier" above) so that its calls allocate nothing after
 * the first one, like the allocation-free reference (lib.rs:3).  It is freed at thread exit; this
 * frees the calling thread's context early (long-lived worker threads should call it before they exit: at process
 * teardown the destructor frees nothing if the HIP runtime no longer answers). */
void fl_host_release(void);
/* (The kernel-selection override used by the parity tests and the A/B tools is NOT part of this interface:
 * include/fastlanes_amd_internal.h.) */

/* Test / benchmark data generated in HBM (not a reference function; SURVEY.md 8(d) asks for a counter-based generator
 * so that a host can regenerate any part of a device-resident column without a PCIe transfer): 64-bit word i of dst =
 * output number i+1 of splitmix64 seeded with seed * 0x9E3779B97F4A7C15.  dst 8-byte aligned, n_bytes a multiple of 8
 * (FL_ERR_ALIGN otherwise).  Asynchronous on `stream`. */
int fl_fill_random(void *dst, size_t n_bytes, uint64_t seed, void *stream);

/*
 * OPTIONAL allocation helper -- never needed to use the codec: every entry point takes any 16-byte aligned device pointers and
 * allocates nothing.  It exists because WHERE a column's buffers live in HBM moves every streaming kernel by a few per cent on
 * MI355X (the same kernel, the same bytes, another allocation: 0.78-0.86 of the peak), nothing in an address
 * tells which, and a caller that allocates a (packed, unpacked) pair it is going to stream through many times can MEASURE it once:
 *   FL_LAYOUT_SEPARATE     one hipMalloc per buffer, wherever the driver puts them
 *   FL_LAYOUT_ZONED        one hipMalloc; `in` (then `aux`) at its start, `out` centred on the first 64-GiB multiple that leaves room
 *                          for them.  COSTS THE UNUSED BYTES IN BETWEEN: the allocation is about 64 GiB + out_bytes / 2 whatever the
 *                          pair's size (a multiple of 64 GiB more for inputs beyond ~60 GiB) -- a few such pairs exhaust a device.
 *   FL_LAYOUT_INTERLEAVED  CONSTRUCTED (round 6; DESIGN.md section 4, profiles/r06_vmm_placement.txt): the pair is built from 1-GiB
 *                          physical chunks (hipMemCreate) mapped into ONE address range (hipMemAddressReserve / hipMemMap).  The device
 *                          memory has three classes (most likely the three ranks of the HBM stacks; nothing in an address tells which):
 *                          concurrent writes are fastest spread over two classes, reads inside one, and reads and writes in different
 *                          ones.  More chunks than needed are created, every chunk's class is MEASURED (a 0.2-ms probe kernel per chunk and
 *                          class), `in` (and `aux`) get chunks of one class, `out`'s chunks are arranged so that the eight XCDs' concurrent write
 *                          positions (XCD x walks the x-th eighth of `out`) are spread evenly over classes at every moment -- over the
 *                          two classes `in` is not in when out_bytes >= 3 * in_bytes, over all three otherwise -- the rest is
 *                          released.  While a pair is alive, calls whose buffers lie inside it launch under the whole-column tile map.
 *                          u32 W=7 unpack at 10 M blocks: 0.866-0.869 of the 8 TB/s, pack 0.854, where two hipMallocs give anything
 *                          from 0.78 (both buffers in one class) to 0.86.  Transient cost: a pool of twice the pair (at least 48 GiB, growing to three
 *                          times the pair where the first pool lacks a class) for a second or so; pairs below 8 GiB are allocated as FL_LAYOUT_SEPARATE (a handful of chunks: nothing to arrange);
 *                          FL_ERR_HIP with hipErrorNotSupported where the device has no virtual-memory management.
 *   FL_LAYOUT_PROBE        the candidates are allocated one after the other -- INTERLEAVED, SEPARATE, and ZONED where its slab still fits --
 *                          a bare read/write stream of in_bytes : out_bytes (no codec work; fl_stream.hpp) is timed on each for a few
 *                          launches on `stream`, the fastest pair is kept (a later candidate must win by more than 1 %, ZONED by more
 *                          than 2 %: it pins ~64 GiB), the others are freed.  SYNCHRONOUS (about 10 launches over the buffers per
 *                          candidate; `stream` must not be capturing), and the buffers' contents are unspecified afterwards.  A candidate
 *                          that cannot be allocated is skipped; pairs too small to time (< 4 MiB) are allocated SEPARATE.
 * in / aux / out receive in_bytes / aux_bytes / out_bytes bytes (256-byte aligned; aux_bytes may be 0: *aux = NULL, aux itself may then
 * be NULL); *handle owns the memory: fl_column_pair_free(handle) releases it (NULL is a no-op).  layout_kept (may be NULL) receives
 * the layout of the returned pair, probe_gbps (may be NULL; FL_LAYOUT_COUNT entries, indexed by layout, the FL_LAYOUT_PROBE slot stays
 * 0) the probe's GB/s (0 = not measured).  This is what bench.py's --placement auto does: the figure it prints is one this header
 * alone reproduces.
 * (An INTERLEAVED pair's addresses are never handed out twice within a process: on this ROCm (7.2) an address range that is unmapped and
 * mapped again keeps translating to the FIRST chunks it held -- tools/exp_vmm remap -- so the library takes its ranges from a private,
 * monotonically growing part of the address space, 16 .. 112 TiB: a pair uses its own size plus its pool's there, once, so a process can
 * construct a few hundred large pairs; after that INTERLEAVED fails with FL_ERR_HIP (hipErrorOutOfMemory) and PROBE falls back to the plain
 * layouts.)
 */
enum { FL_LAYOUT_SEPARATE = 0, FL_LAYOUT_ZONED = 1, FL_LAYOUT_PROBE = 2, FL_LAYOUT_INTERLEAVED = 3, FL_LAYOUT_COUNT = 4 };
int fl_column_pair_alloc(size_t in_bytes, size_t aux_bytes, size_t out_bytes, int layout, void *stream,
                         void **in, void **aux, void **out, void **handle, int *layout_kept,
                         uint32_t *probe_gbps);
int fl_column_pair_free(void *handle);

/*
 * Mixed-width columns (BASELINE.json config 5): block b has its own width widths[b].
 * The reference has no multi-block API; this is its caller loop
 *     for b in blocks { T::unchecked_unpack(widths[b], &packed[off[b]..], &mut out[b*1024..]) }
 * (bitpacking.rs:109-129; pack: bitpacking.rs:76-96; loop shape of benches/bitpacking.rs:80-97)
 * moved on-device.  The surface is the one SURVEY.md 8(b) names: two DEVICE arrays,
 *     widths [n_blocks]  uint8   width of block b (<= T)
 *     offsets[n_blocks]  uint64  byte offset of block b's 128*widths[b] bytes in the packed column
 *                                (multiples of 16; back-to-back blocks give multiples of 128)
 * read by the kernel itself (fl_<ty>_unpack_widths / fl_<ty>_pack_widths below): one wavefront per
 * block, blocks in column order, ONE launch, no host pass over the column, no allocation, any
 * block count.  `packed_bytes` is the size of the packed column.  The kernel checks every block's
 * preconditions itself: a block whose width exceeds T, whose offset is not a multiple of 16, or whose
 * 128*widths[b] bytes do not lie inside [0, packed_bytes) is SKIPPED (nothing is read or written for
 * it) and its FL_DEVERR_* bit is ORed into *err_flag -- the device-side form of bitpacking.rs:93/126
 * unreachable!() and of the length debug_asserts (:78-80, :111-113).
 *
 * fl_widths_to_offsets builds the back-to-back offsets on the device: offsets[b] = sum_{i<b}
 * 128*widths[i] (exclusive prefix sum, three small launches on `stream`, no scratch memory),
 * *total_bytes (device uint64, may be NULL) = the packed column's size, FL_DEVERR_WIDTH ORed into
 * *err_flag if some width exceeds type_bits.
 */
int fl_widths_to_offsets(unsigned type_bits, const uint8_t *widths, size_t n_blocks,
                         uint64_t *offsets, uint64_t *total_bytes, uint32_t *err_flag,
                         void *stream);

/*
 * Convenience owner of the two device arrays for callers that hold the widths on the HOST:
 * create validates the widths (FL_ERR_WIDTH), uploads them to the current device, runs
 * fl_widths_to_offsets and reads back the packed size.  fl_<ty>_unpack_mixed / pack_mixed are
 * fl_<ty>_unpack_widths / pack_widths over the plan's arrays.  The plan must be used on the
 * device it was created on and destroyed by the caller.
 */
typedef struct fl_mixed_plan fl_mixed_plan;
int fl_mixed_plan_create(unsigned type_bits, const uint8_t *widths, size_t n_blocks,
                         fl_mixed_plan **plan);
void fl_mixed_plan_destroy(fl_mixed_plan *plan);
size_t fl_mixed_plan_n_blocks(const fl_mixed_plan *plan);
/* total packed bytes of the column = sum 128*widths[b] */
uint64_t fl_mixed_plan_packed_bytes(const fl_mixed_plan *plan);
/* device pointers to the plan's uint64 byte offsets / uint8 widths [n_blocks] (owned by the plan) */
const uint64_t *fl_mixed_plan_offsets(const fl_mixed_plan *plan);
const uint8_t *fl_mixed_plan_widths(const fl_mixed_plan *plan);

#define FL_DECLARE_TYPE(T, S)                                                                   \
    /* BitPacking::unchecked_pack (bitpacking.rs:30,76-96) -> pack::<W> (:65-74) */             \
    int fl_##S##_pack(unsigned width, const T *in, T *out, size_t n_blocks, void *stream);      \
    /* BitPacking::unchecked_unpack (bitpacking.rs:44,109-129) -> unpack::<W> (:98-107) */      \
    int fl_##S##_unpack(unsigned width, const T *in, T *out, size_t n_blocks, void *stream);    \
    /* BitPacking::unchecked_unpack_single (bitpacking.rs:58,181-200) -> unpack_single::<W>    \
     * (:132-179), batched: out[k] = value at element indices[k] of the column, where          \
     * indices[k] = block*1024 + index_in_block.  Out-of-range indices make the call return    \
     * FL_ERR_INDEX on the host tier; on the device tier they write 0 and OR FL_DEVERR_INDEX   \
     * into *err_flag (a device uint32, may be NULL). */                                       \
    int fl_##S##_unpack_single(unsigned width, const T *packed, size_t n_blocks,                \
                               const uint64_t *indices, size_t n_indices, T *out,               \
                               uint32_t *err_flag, void *stream);                               \
    /* FoR::for_pack::<W> (ffor.rs:5-9,24-36).  references[b*reference_stride] is block b's    \
     * scalar; reference_stride 0 broadcasts references[0]. */                                 \
    int fl_##S##_for_pack(unsigned width, const T *in, const T *references,                     \
                          size_t reference_stride, T *out, size_t n_blocks, void *stream);      \
    /* FoR::unfor_pack::<W> (ffor.rs:11-17,38-50) */                                            \
    int fl_##S##_unfor_pack(unsigned width, const T *in, const T *references,                   \
                            size_t reference_stride, T *out, size_t n_blocks, void *stream);    \
    /* Delta::delta (delta.rs:7,24-33); bases is [n_blocks][LANES] (128 bytes per block) */     \
    int fl_##S##_delta(const T *in, const T *bases, T *out, size_t n_blocks, void *stream);     \
    /* Delta::undelta (delta.rs:9,36-45) */                                                     \
    int fl_##S##_undelta(const T *in, const T *bases, T *out, size_t n_blocks, void *stream);   \
    /* Delta::undelta_pack::<W> (delta.rs:11-16,47-63); output stays in transposed order */     \
    int fl_##S##_undelta_pack(unsigned width, const T *in, const T *bases, T *out,              \
                              size_t n_blocks, void *stream);                                   \
    /* EXTENSIONS beyond the reference's methods; defined as the compositions  \
     * the reference's own test/bench perform (delta.rs:88-100, benches/delta.rs:20-27):               \
     *   undelta_pack_untranspose(W, pk, bases) == untranspose(undelta_pack::<W>(pk, bases))           \
     *   transpose_delta_pack(W, v, bases)      == pack::<W>(delta(transpose(v), bases))               \
     * i.e. decode straight to ORIGINAL order / encode straight from it, saving one 2x(1024*T/8)-byte  \
     * round trip per block. */                                                                       \
    int fl_##S##_undelta_pack_untranspose(unsigned width, const T *in, const T *bases, T *out,  \
                                          size_t n_blocks, void *stream);                       \
    int fl_##S##_transpose_delta_pack(unsigned width, const T *in, const T *bases, T *out,      \
                                      size_t n_blocks, void *stream);                           \
    /* EXTENSIONS, reductions over what the reference functions produce:           \
     *   unpack_block_sums: sums[b] = sum of the 1024 values unpack::<W> yields for block b           \
     *                      (wrapping uint64) without materialising them: 128*W bytes in, 8 out.       \
     *   block_min_max:     mins[b], maxs[b] over unpacked block b -- an encoder's inputs for FoR's    \
     *                      reference and width before for_pack::<W> (ffor.rs:24-36). */              \
    int fl_##S##_unpack_block_sums(unsigned width, const T *in, size_t n_blocks, uint64_t *sums,  \
                                   void *stream);                                               \
    int fl_##S##_block_min_max(const T *in, size_t n_blocks, T *mins, T *maxs, void *stream);    \
    /*   unpack_compare:    bit i of mask[b*32 .. b*32+32) = (unpack::<W>(block b)[i] <op> constant),  \
     *                      i in the unpacked (index) order: a selection vector straight from packed   \
     *                      data, 128*W bytes in, 128 bytes out per block.  op is an fl_cmp (any other    \
     *                      value: FL_ERR_INDEX). */       \
    int fl_##S##_unpack_compare(unsigned width, const T *in, int op, T constant, size_t n_blocks, \
                                uint32_t *mask, void *stream);                                  \
    /* Transpose::transpose (transpose.rs:5,11-15) */                                           \
    int fl_##S##_transpose(const T *in, T *out, size_t n_blocks, void *stream);                 \
    /* Transpose::untranspose (transpose.rs:6,17-22) */                                         \
    int fl_##S##_untranspose(const T *in, T *out, size_t n_blocks, void *stream);               \
    /* unchecked_unpack / unchecked_pack (bitpacking.rs:109-129, :76-96) looped over blocks with    \
     * per-block device widths[] / offsets[] (see "Mixed-width columns" above) */                  \
    int fl_##S##_unpack_widths(const uint8_t *widths, const uint64_t *offsets, const T *packed,  \
                               size_t packed_bytes, T *out, size_t n_blocks, uint32_t *err_flag,  \
                               void *stream);                                                    \
    int fl_##S##_pack_widths(const uint8_t *widths, const uint64_t *offsets, const T *in,        \
                             T *packed, size_t packed_bytes, size_t n_blocks, uint32_t *err_flag, \
                             void *stream);                                                      \
    /* FoR's and Delta's bodies over such a column -- the reference's const-W methods called with block b's width, as its     \
     * callers do per chunk:  unfor_pack::<widths[b]> / for_pack::<widths[b]> (ffor.rs:24-50) with references[b*reference_stride] \
     * (stride 0 broadcasts references[0]);  undelta_pack::<widths[b]> (delta.rs:47-63) with bases[b][LANES], its output in   \
     * transposed order, and the two fused transpose extensions above.  Same per-block checks, same *err_flag. */             \
    int fl_##S##_unfor_pack_widths(const uint8_t *widths, const uint64_t *offsets, const T *packed,  \
                                   size_t packed_bytes, const T *references, size_t reference_stride, \
                                   T *out, size_t n_blocks, uint32_t *err_flag, void *stream);        \
    int fl_##S##_for_pack_widths(const uint8_t *widths, const uint64_t *offsets, const T *in,         \
                                 const T *references, size_t reference_stride, T *packed,             \
                                 size_t packed_bytes, size_t n_blocks, uint32_t *err_flag,            \
                                 void *stream);                                                       \
    int fl_##S##_undelta_pack_widths(const uint8_t *widths, const uint64_t *offsets, const T *packed, \
                                     size_t packed_bytes, const T *bases, T *out, size_t n_blocks,    \
                                     uint32_t *err_flag, void *stream);                               \
    int fl_##S##_undelta_pack_untranspose_widths(const uint8_t *widths, const uint64_t *offsets,      \
                                                 const T *packed, size_t packed_bytes, const T *bases, \
                                                 T *out, size_t n_blocks, uint32_t *err_flag,         \
                                                 void *stream);                                       \
    int fl_##S##_transpose_delta_pack_widths(const uint8_t *widths, const uint64_t *offsets,          \
                                             const T *in, const T *bases, T *packed,                  \
                                             size_t packed_bytes, size_t n_blocks, uint32_t *err_flag, \
                                             void *stream);                                           \
    /* EXTENSION, the step between block_min_max and for_pack_widths in an encoder:  widths[b] = number of  \
     * bits of maxs[b] - mins[b] (0 for a constant block) -- the smallest W for which for_pack::<W>(block b, mins[b])         \
     * (ffor.rs:24-36; masked to W bits by macros.rs:73) loses nothing.  The reference itself selects no widths. */           \
    int fl_##S##_for_widths(const T *mins, const T *maxs, size_t n_blocks, uint8_t *widths,          \
                            void *stream);                                                            \
    /* unchecked_unpack_single (bitpacking.rs:58,181-200) over such a column, batched: out[k] = element  \
     * indices[k] (= block*1024 + index_in_block) of the column; an index past the column, a width  \
     * > T or a block outside the packed column writes 0 and ORs its FL_DEVERR_* bit into *err_flag */ \
    int fl_##S##_unpack_single_widths(const uint8_t *widths, const uint64_t *offsets,            \
                                      const T *packed, size_t packed_bytes, size_t n_blocks,      \
                                      const uint64_t *indices, size_t n_indices, T *out,          \
                                      uint32_t *err_flag, void *stream);                          \
    /* MANY SMALL ARRAYS in one launch -- the shape of a columnar engine's chunks (Vortex: 64 Ki values = 64 blocks per   \
     * chunk), whose per-chunk loop over unchecked_unpack / unchecked_pack (bitpacking.rs:109-129, :76-96) is launch-bound  \
     * when every chunk is its own call.  Four DEVICE arrays of length n_arrays: packed[a] / out[a] are device pointers      \
     * (16-byte aligned) to array a's packed and unpacked blocks, widths[a] its width, n_blocks[a] its block count;          \
     * max_blocks (host) >= every n_blocks[a] sizes the grid.  An array with a width > T or a misaligned / NULL pointer is    \
     * skipped and FL_DEVERR_WIDTH / FL_DEVERR_ALIGN is ORed into *err_flag; an array with n_blocks[a] > max_blocks has only  \
     * its first max_blocks (rounded up to a workgroup's share, 4 to 64 blocks) blocks processed and raises FL_DEVERR_BOUNDS;     \
     * max_blocks itself may not exceed 2^30 (FL_ERR_INDEX). */                                \
    int fl_##S##_unpack_batch(const T *const *packed, T *const *out, const uint8_t *widths,       \
                              const uint32_t *n_blocks, size_t n_arrays, uint32_t max_blocks,     \
                              uint32_t *err_flag, void *stream);                                  \
    int fl_##S##_pack_batch(const T *const *in, T *const *packed, const uint8_t *widths,          \
                            const uint32_t *n_blocks, size_t n_arrays, uint32_t max_blocks,       \
                            uint32_t *err_flag, void *stream);                                    \
    /* ... with FoR's bodies (ffor.rs:24-50): references[a] (a DEVICE array, one scalar per ARRAY -- a chunk's frame of       \
     * reference) is added to / subtracted from every value of array a: unfor_pack::<W> / for_pack::<W> per block. */         \
    int fl_##S##_unfor_pack_batch(const T *const *packed, T *const *out, const uint8_t *widths,   \
                                  const T *references, const uint32_t *n_blocks, size_t n_arrays, \
                                  uint32_t max_blocks, uint32_t *err_flag, void *stream);         \
    int fl_##S##_for_pack_batch(const T *const *in, T *const *packed, const uint8_t *widths,      \
                                const T *references, const uint32_t *n_blocks, size_t n_arrays,   \
                                uint32_t max_blocks, uint32_t *err_flag, void *stream);           \
    /* ... with Delta's bodies: bases[a] is a device pointer to array a's bases, [n_blocks[a]][LANES] (128 bytes per block):   \
     * undelta_pack::<W> per block (delta.rs:47-63; output in transposed order, or -- untranspose != 0 -- straight in original  \
     * order: the fused extension above), and the fused encode pack::<W>(delta(transpose(in), bases)). */                       \
    int fl_##S##_undelta_pack_batch(const T *const *packed, const T *const *bases, T *const *out,  \
                                    const uint8_t *widths, const uint32_t *n_blocks, size_t n_arrays, \
                                    uint32_t max_blocks, int untranspose, uint32_t *err_flag,       \
                                    void *stream);                                                  \
    int fl_##S##_transpose_delta_pack_batch(const T *const *in, const T *const *bases,             \
                                            T *const *packed, const uint8_t *widths,                \
                                            const uint32_t *n_blocks, size_t n_arrays,              \
                                            uint32_t max_blocks, uint32_t *err_flag, void *stream); \
    /* the same over a mixed-width plan (see fl_mixed_plan) */                                     \
    int fl_##S##_unpack_mixed(const fl_mixed_plan *plan, const T *packed, T *out, void *stream); \
    int fl_##S##_pack_mixed(const fl_mixed_plan *plan, const T *in, T *packed, void *stream);    \
    /* ---- host-pointer tier: the trait methods' own slice arguments ---- */                   \
    int fl_##S##_pack_host(unsigned width, const T *in, T *out, size_t n_blocks);               \
    int fl_##S##_unpack_host(unsigned width, const T *in, T *out, size_t n_blocks);             \
    int fl_##S##_unpack_single_host(unsigned width, const T *packed, size_t n_blocks,           \
                                    uint64_t index, T *value);                                  \
    int fl_##S##_for_pack_host(unsigned width, const T *in, T reference, T *out,                \
                               size_t n_blocks);                                                \
    int fl_##S##_unfor_pack_host(unsigned width, const T *in, T reference, T *out,              \
                                 size_t n_blocks);                                              \
    int fl_##S##_delta_host(const T *in, const T *bases, T *out, size_t n_blocks);              \
    int fl_##S##_undelta_host(const T *in, const T *bases, T *out, size_t n_blocks);            \
    int fl_##S##_undelta_pack_host(unsigned width, const T *in, const T *bases, T *out,         \
                                   size_t n_blocks);                                            \
    int fl_##S##_transpose_host(const T *in, T *out, size_t n_blocks);                          \
    int fl_##S##_untranspose_host(const T *in, T *out, size_t n_blocks);

FL_DECLARE_TYPE(uint8_t, u8)
FL_DECLARE_TYPE(uint16_t, u16)
FL_DECLARE_TYPE(uint32_t, u32)
FL_DECLARE_TYPE(uint64_t, u64)

#ifdef __cplusplus
}
#endif
#endif /* FASTLANES_AMD_H */
