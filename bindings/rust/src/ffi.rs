//! Raw declarations of the C ABI in `include/fastlanes_amd.h` (host tier + the batched device
//! tier).  Uncompiled: see ../README.md.
#![allow(dead_code)]
use core::ffi::{c_char, c_void};

macro_rules! declare_type {
    ($T:ty, $pack_host:ident, $unpack_host:ident, $single_host:ident, $for_pack_host:ident, $unfor_pack_host:ident,
     $delta_host:ident, $undelta_host:ident, $undelta_pack_host:ident, $transpose_host:ident, $untranspose_host:ident,
     $pack:ident, $unpack:ident, $for_pack:ident, $unfor_pack:ident, $delta:ident, $undelta:ident,
     $undelta_pack:ident, $transpose:ident, $untranspose:ident) => {
        extern "C" {
            // host tier: the trait methods' own slices; n_blocks = 1 per trait call
            pub fn $pack_host(width: u32, input: *const $T, output: *mut $T, n_blocks: usize) -> i32;
            pub fn $unpack_host(width: u32, input: *const $T, output: *mut $T, n_blocks: usize) -> i32;
            pub fn $single_host(width: u32, packed: *const $T, n_blocks: usize, index: u64, value: *mut $T) -> i32;
            pub fn $for_pack_host(width: u32, input: *const $T, reference: $T, output: *mut $T, n_blocks: usize) -> i32;
            pub fn $unfor_pack_host(width: u32, input: *const $T, reference: $T, output: *mut $T, n_blocks: usize) -> i32;
            pub fn $delta_host(input: *const $T, bases: *const $T, output: *mut $T, n_blocks: usize) -> i32;
            pub fn $undelta_host(input: *const $T, bases: *const $T, output: *mut $T, n_blocks: usize) -> i32;
            pub fn $undelta_pack_host(width: u32, input: *const $T, bases: *const $T, output: *mut $T, n_blocks: usize) -> i32;
            pub fn $transpose_host(input: *const $T, output: *mut $T, n_blocks: usize) -> i32;
            pub fn $untranspose_host(input: *const $T, output: *mut $T, n_blocks: usize) -> i32;
            // device tier: device pointers, n_blocks contiguous blocks, asynchronous on a hipStream_t
            pub fn $pack(width: u32, d_in: *const $T, d_out: *mut $T, n_blocks: usize, stream: *mut c_void) -> i32;
            pub fn $unpack(width: u32, d_in: *const $T, d_out: *mut $T, n_blocks: usize, stream: *mut c_void) -> i32;
            pub fn $for_pack(width: u32, d_in: *const $T, d_refs: *const $T, ref_stride: usize, d_out: *mut $T,
                             n_blocks: usize, stream: *mut c_void) -> i32;
            pub fn $unfor_pack(width: u32, d_in: *const $T, d_refs: *const $T, ref_stride: usize, d_out: *mut $T,
                               n_blocks: usize, stream: *mut c_void) -> i32;
            pub fn $delta(d_in: *const $T, d_bases: *const $T, d_out: *mut $T, n_blocks: usize, stream: *mut c_void) -> i32;
            pub fn $undelta(d_in: *const $T, d_bases: *const $T, d_out: *mut $T, n_blocks: usize, stream: *mut c_void) -> i32;
            pub fn $undelta_pack(width: u32, d_in: *const $T, d_bases: *const $T, d_out: *mut $T, n_blocks: usize,
                                 stream: *mut c_void) -> i32;
            pub fn $transpose(d_in: *const $T, d_out: *mut $T, n_blocks: usize, stream: *mut c_void) -> i32;
            pub fn $untranspose(d_in: *const $T, d_out: *mut $T, n_blocks: usize, stream: *mut c_void) -> i32;
        }
    };
}

declare_type!(u8, fl_u8_pack_host, fl_u8_unpack_host, fl_u8_unpack_single_host, fl_u8_for_pack_host, fl_u8_unfor_pack_host,
              fl_u8_delta_host, fl_u8_undelta_host, fl_u8_undelta_pack_host, fl_u8_transpose_host, fl_u8_untranspose_host,
              fl_u8_pack, fl_u8_unpack, fl_u8_for_pack, fl_u8_unfor_pack, fl_u8_delta, fl_u8_undelta,
              fl_u8_undelta_pack, fl_u8_transpose, fl_u8_untranspose);
declare_type!(u16, fl_u16_pack_host, fl_u16_unpack_host, fl_u16_unpack_single_host, fl_u16_for_pack_host, fl_u16_unfor_pack_host,
              fl_u16_delta_host, fl_u16_undelta_host, fl_u16_undelta_pack_host, fl_u16_transpose_host, fl_u16_untranspose_host,
              fl_u16_pack, fl_u16_unpack, fl_u16_for_pack, fl_u16_unfor_pack, fl_u16_delta, fl_u16_undelta,
              fl_u16_undelta_pack, fl_u16_transpose, fl_u16_untranspose);
declare_type!(u32, fl_u32_pack_host, fl_u32_unpack_host, fl_u32_unpack_single_host, fl_u32_for_pack_host, fl_u32_unfor_pack_host,
              fl_u32_delta_host, fl_u32_undelta_host, fl_u32_undelta_pack_host, fl_u32_transpose_host, fl_u32_untranspose_host,
              fl_u32_pack, fl_u32_unpack, fl_u32_for_pack, fl_u32_unfor_pack, fl_u32_delta, fl_u32_undelta,
              fl_u32_undelta_pack, fl_u32_transpose, fl_u32_untranspose);
declare_type!(u64, fl_u64_pack_host, fl_u64_unpack_host, fl_u64_unpack_single_host, fl_u64_for_pack_host, fl_u64_unfor_pack_host,
              fl_u64_delta_host, fl_u64_undelta_host, fl_u64_undelta_pack_host, fl_u64_transpose_host, fl_u64_untranspose_host,
              fl_u64_pack, fl_u64_unpack, fl_u64_for_pack, fl_u64_unfor_pack, fl_u64_delta, fl_u64_undelta,
              fl_u64_undelta_pack, fl_u64_transpose, fl_u64_untranspose);

// Mixed-width columns: the caller loop `for b { T::unchecked_unpack(widths[b], &packed[off[b]..], ..) }`
// (bitpacking.rs:109-129) with widths[] (u8) and offsets[] (u64 byte offsets) resident in HBM.
macro_rules! declare_widths {
    ($T:ty, $unpack_widths:ident, $pack_widths:ident, $unpack_single_widths:ident) => {
        extern "C" {
            pub fn $unpack_single_widths(d_widths: *const u8, d_offsets: *const u64, d_packed: *const $T, packed_bytes: usize, n_blocks: usize,
                                         d_indices: *const u64, n_indices: usize, d_out: *mut $T, d_err_flag: *mut u32,
                                         stream: *mut c_void) -> i32;
            pub fn $unpack_widths(d_widths: *const u8, d_offsets: *const u64, d_packed: *const $T, packed_bytes: usize, d_out: *mut $T,
                                  n_blocks: usize, d_err_flag: *mut u32, stream: *mut c_void) -> i32;
            pub fn $pack_widths(d_widths: *const u8, d_offsets: *const u64, d_in: *const $T, d_packed: *mut $T, packed_bytes: usize,
                                n_blocks: usize, d_err_flag: *mut u32, stream: *mut c_void) -> i32;
        }
    };
}
declare_widths!(u8, fl_u8_unpack_widths, fl_u8_pack_widths, fl_u8_unpack_single_widths);
declare_widths!(u16, fl_u16_unpack_widths, fl_u16_pack_widths, fl_u16_unpack_single_widths);
declare_widths!(u32, fl_u32_unpack_widths, fl_u32_pack_widths, fl_u32_unpack_single_widths);
declare_widths!(u64, fl_u64_unpack_widths, fl_u64_pack_widths, fl_u64_unpack_single_widths);

extern "C" {
    /// offsets[b] = sum_{i<b} 128 * widths[i] on the device (three small launches, no scratch memory)
    pub fn fl_widths_to_offsets(type_bits: u32, d_widths: *const u8, n_blocks: usize, d_offsets: *mut u64,
                                d_total_bytes: *mut u64, d_err_flag: *mut u32, stream: *mut c_void) -> i32;
    /// classes[g] = memory class (0, 1, 2; -1 = no clean answer) of every 8-GiB granule of a device allocation: decode outputs
    /// are fastest across a class boundary, a mask or per-block sums fastest in another class than the packed input
    /// (fastlanes_amd.h; synchronous and destructive: call it on a pool before the pool holds data)
    pub fn fl_probe_memory_classes(slab: *mut c_void, slab_bytes: usize, classes: *mut i32, stream: *mut c_void) -> i32;
    /// frees the calling thread's cached host-tier context (stream, pinned + device staging buffers)
    pub fn fl_host_release();
    pub fn fl_status_string(status: i32) -> *const c_char;
    pub fn fl_last_hip_error() -> i32;
}

/// The reference has no `Result`: `width > T` is `unreachable!()` (bitpacking.rs:93,126,197) and an
/// out-of-range index is `assert!` (bitpacking.rs:152).  A nonzero status therefore panics, as there.
#[inline]
pub(crate) fn check(status: i32, what: &str) {
    if status != 0 {
        panic!("{what}: fastlanes_amd status {status} (hipError_t {})", unsafe { fl_last_hip_error() });
    }
}
