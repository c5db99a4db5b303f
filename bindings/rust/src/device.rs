//! The reference's CALLER LOOP as one checked, device-resident call.
//!
//! Vortex (and `benches/bitpacking.rs:80-97`) hold a column as slices and loop
//! ```text
//! for i in 0..n { T::unchecked_unpack(w, &packed[i*pl..(i+1)*pl], &mut out[i*1024..(i+1)*1024]) }
//! ```
//! where every iteration asserts its two lengths (`bitpacking.rs:78-80`, `:111-113`).  Moved to the GPU that loop is ONE
//! asynchronous launch (`ffi::fl_<ty>_unpack`), and the raw FFI would lose exactly those asserts.  A [`DeviceSlice`] is the
//! device-resident counterpart of `&[T]` -- pointer AND element count, borrow-checked like a slice but never dereferenced on
//! the host -- and the `*_column` methods of [`DeviceCodec`] check the same lengths for the whole column before the launch.
//! The only `unsafe` a caller writes is the construction of a slice from a raw device pointer.
//!
//! Errors follow the reference: a length mismatch is an `assert!` (its `debug_assert!`s, always on here: one check per COLUMN
//! costs nothing), `width > T` panics through `ffi::check` (its `unreachable!()`, `bitpacking.rs:93,126`).
//!
//! `include/fastlanes_amd.hpp` holds the same surface in C++ (`fastlanes::DeviceSlice`, `unpack_column`, ...), which IS
//! compiled and run on the GPU by `tests/cpp/test_trait_mirror.cpp`.  Uncompiled here: see ../README.md.
use core::ffi::c_void;
use core::marker::PhantomData;
use core::mem::size_of;

use crate::ffi;

/// A `hipStream_t`; `Stream::DEFAULT` is the null stream.  Must belong to the calling thread's current device.
#[derive(Clone, Copy)]
pub struct Stream(pub *mut c_void);
impl Stream {
    pub const DEFAULT: Stream = Stream(core::ptr::null_mut());
}

/// `&[T]` in device memory: `len` elements at a 16-byte aligned device pointer of the calling thread's current device.
#[derive(Clone, Copy)]
pub struct DeviceSlice<'a, T> {
    ptr: *const T,
    len: usize,
    _borrow: PhantomData<&'a [T]>,
}
/// `&mut [T]` in device memory.
pub struct DeviceSliceMut<'a, T> {
    ptr: *mut T,
    len: usize,
    _borrow: PhantomData<&'a mut [T]>,
}

impl<'a, T> DeviceSlice<'a, T> {
    /// # Safety
    /// `ptr .. ptr + len` is device memory the current device can read for `'a`, 16-byte aligned.
    pub unsafe fn from_raw_parts(ptr: *const T, len: usize) -> Self {
        Self { ptr, len, _borrow: PhantomData }
    }
    pub fn len(&self) -> usize { self.len }
    pub fn is_empty(&self) -> bool { self.len == 0 }
    pub fn as_ptr(&self) -> *const T { self.ptr }
    /// elements `[first, first + count)`: what a sharded caller hands each device / stream
    pub fn subslice(&self, first: usize, count: usize) -> DeviceSlice<'a, T> {
        assert!(first <= self.len && count <= self.len - first, "DeviceSlice::subslice out of range");
        Self { ptr: self.ptr.wrapping_add(first), len: count, _borrow: PhantomData }
    }
}
impl<'a, T> DeviceSliceMut<'a, T> {
    /// # Safety
    /// `ptr .. ptr + len` is device memory the current device can write for `'a`, 16-byte aligned, aliased by nothing else
    /// that is in use on the stream.
    pub unsafe fn from_raw_parts(ptr: *mut T, len: usize) -> Self {
        Self { ptr, len, _borrow: PhantomData }
    }
    pub fn len(&self) -> usize { self.len }
    pub fn is_empty(&self) -> bool { self.len == 0 }
    pub fn as_mut_ptr(&mut self) -> *mut T { self.ptr }
    pub fn as_slice(&self) -> DeviceSlice<'_, T> {
        DeviceSlice { ptr: self.ptr, len: self.len, _borrow: PhantomData }
    }
    pub fn subslice_mut(&mut self, first: usize, count: usize) -> DeviceSliceMut<'_, T> {
        assert!(first <= self.len && count <= self.len - first, "DeviceSliceMut::subslice_mut out of range");
        DeviceSliceMut { ptr: self.ptr.wrapping_add(first), len: count, _borrow: PhantomData }
    }
}

/// Number of 1024-value blocks of a column given as (packed, unpacked) lengths; panics unless `unpacked == 1024 * n` and
/// `packed == n * 1024 * width / T` for one `n` -- `bitpacking.rs:78-80` / `:111-113`, for every block of the column at once.
fn column_blocks<T>(width: usize, packed_len: usize, unpacked_len: usize, what: &str) -> usize {
    let t_bits = 8 * size_of::<T>();
    assert!(width <= t_bits, "{what}: width {width} > T = {t_bits}");                      // bitpacking.rs:93,126 unreachable!()
    assert_eq!(unpacked_len % 1024, 0, "{what}: the unpacked slice must hold 1024 elements per block");
    let n = unpacked_len / 1024;
    assert_eq!(packed_len, n * (128 * width / size_of::<T>()), "{what}: the packed slice must hold 1024 * W / T elements per block");
    n
}

/// The four per-chunk device arrays of `fl_<ty>_unpack_batch` (many small arrays -- Vortex's 64 Ki-value chunks -- in ONE launch)
/// with their lengths.  What only the device can see (each chunk's width, the pointers' alignment, its block count against
/// `max_blocks`) is checked by the kernel and reported through `*err_flag` (`ffi::FL_DEVERR_*`).
pub struct ChunkTable<'a, T> {
    pub packed: DeviceSlice<'a, *const T>,
    pub out: DeviceSlice<'a, *mut T>,
    pub widths: DeviceSlice<'a, u8>,
    pub n_blocks: DeviceSlice<'a, u32>,
    /// host-side bound on `n_blocks[c]`; sizes the grid
    pub max_blocks: u32,
}

/// Batched, device-resident forms of the codec traits' methods: the caller loop as one checked call.
pub trait DeviceCodec: Sized {
    /// `for b in blocks { Self::unchecked_unpack(width, &packed[b*pl..], &mut out[b*1024..]) }` (bitpacking.rs:109-129)
    fn unpack_column(width: usize, packed: DeviceSlice<Self>, out: &mut DeviceSliceMut<Self>, stream: Stream);
    /// `for b in blocks { Self::unchecked_pack(width, &input[b*1024..], &mut packed[b*pl..]) }` (bitpacking.rs:76-96)
    fn pack_column(width: usize, input: DeviceSlice<Self>, packed: &mut DeviceSliceMut<Self>, stream: Stream);
    /// `for b in blocks { Self::undelta_pack::<W>(&packed[b], &bases[b], &mut out[b]) }` (delta.rs:47-63); `bases`: LANES per block
    fn undelta_pack_column(width: usize, packed: DeviceSlice<Self>, bases: DeviceSlice<Self>, out: &mut DeviceSliceMut<Self>, stream: Stream);
    /// `for b in blocks { Self::unfor_pack::<W>(&packed[b], references[b], &mut out[b]) }` (ffor.rs:38-50); one reference per
    /// block, or one for the whole column (`references.len() == 1`)
    fn unfor_pack_column(width: usize, packed: DeviceSlice<Self>, references: DeviceSlice<Self>, out: &mut DeviceSliceMut<Self>, stream: Stream);
    /// the same loop over many small arrays, one launch; `err_flag`: one zeroed device `u32`, or empty
    fn unpack_chunks(table: &ChunkTable<Self>, err_flag: &mut DeviceSliceMut<u32>, stream: Stream);
}

macro_rules! impl_device_codec {
    ($T:ty, $unpack:ident, $pack:ident, $undelta_pack:ident, $unfor_pack:ident, $unpack_batch:ident) => {
        impl DeviceCodec for $T {
            fn unpack_column(width: usize, packed: DeviceSlice<Self>, out: &mut DeviceSliceMut<Self>, stream: Stream) {
                let n = column_blocks::<Self>(width, packed.len(), out.len(), "unpack_column");
                ffi::check(unsafe { ffi::$unpack(width as u32, packed.as_ptr(), out.as_mut_ptr(), n, stream.0) }, "unpack_column");
            }
            fn pack_column(width: usize, input: DeviceSlice<Self>, packed: &mut DeviceSliceMut<Self>, stream: Stream) {
                let n = column_blocks::<Self>(width, packed.len(), input.len(), "pack_column");
                ffi::check(unsafe { ffi::$pack(width as u32, input.as_ptr(), packed.as_mut_ptr(), n, stream.0) }, "pack_column");
            }
            fn undelta_pack_column(width: usize, packed: DeviceSlice<Self>, bases: DeviceSlice<Self>, out: &mut DeviceSliceMut<Self>, stream: Stream) {
                let n = column_blocks::<Self>(width, packed.len(), out.len(), "undelta_pack_column");
                assert_eq!(bases.len(), n * (1024 / (8 * size_of::<Self>())), "undelta_pack_column: bases must hold LANES elements per block");
                ffi::check(unsafe { ffi::$undelta_pack(width as u32, packed.as_ptr(), bases.as_ptr(), out.as_mut_ptr(), n, stream.0) },
                           "undelta_pack_column");
            }
            fn unfor_pack_column(width: usize, packed: DeviceSlice<Self>, references: DeviceSlice<Self>, out: &mut DeviceSliceMut<Self>, stream: Stream) {
                let n = column_blocks::<Self>(width, packed.len(), out.len(), "unfor_pack_column");
                assert!(references.len() == n || references.len() == 1, "unfor_pack_column: one reference per block, or one in all");
                let stride = if references.len() == 1 { 0 } else { 1 };
                ffi::check(unsafe { ffi::$unfor_pack(width as u32, packed.as_ptr(), references.as_ptr(), stride, out.as_mut_ptr(), n, stream.0) },
                           "unfor_pack_column");
            }
            fn unpack_chunks(table: &ChunkTable<Self>, err_flag: &mut DeviceSliceMut<u32>, stream: Stream) {
                let n = table.widths.len();
                assert!(table.packed.len() == n && table.out.len() == n && table.n_blocks.len() == n,
                        "unpack_chunks: every array of the table holds one entry per chunk");
                assert!(err_flag.len() <= 1, "unpack_chunks: err_flag is one device u32 (or empty)");
                let ef = if err_flag.is_empty() { core::ptr::null_mut() } else { err_flag.as_mut_ptr() };
                ffi::check(unsafe { ffi::$unpack_batch(table.packed.as_ptr(), table.out.as_ptr(), table.widths.as_ptr(), table.n_blocks.as_ptr(), n,
                                                       table.max_blocks, ef, stream.0) }, "unpack_chunks");
            }
        }
    };
}

impl_device_codec!(u8, fl_u8_unpack, fl_u8_pack, fl_u8_undelta_pack, fl_u8_unfor_pack, fl_u8_unpack_batch);
impl_device_codec!(u16, fl_u16_unpack, fl_u16_pack, fl_u16_undelta_pack, fl_u16_unfor_pack, fl_u16_unpack_batch);
impl_device_codec!(u32, fl_u32_unpack, fl_u32_pack, fl_u32_undelta_pack, fl_u32_unfor_pack, fl_u32_unpack_batch);
impl_device_codec!(u64, fl_u64_unpack, fl_u64_pack, fl_u64_undelta_pack, fl_u64_unfor_pack, fl_u64_unpack_batch);

/// The optional allocation helper of `fastlanes_amd.h` as an owner: a (read side, write side) buffer pair for a column, placed by
/// measurement by default (`ffi::FL_LAYOUT_PROBE`: the library allocates every layout it knows -- the one constructed from measured 1-GiB
/// chunks, `FL_LAYOUT_INTERLEAVED`, first --, times a bare read : write stream on each and keeps the fastest pair -- synchronous, contents
/// unspecified afterwards).  Never needed to use the codec: every call above
/// takes any 16-byte aligned device pointers.  (`fastlanes::ColumnPair` in `include/fastlanes_amd.hpp` is the compiled, GPU-tested twin.)
pub struct ColumnPair<T> {
    handle: *mut c_void,
    input: *mut T,
    output: *mut T,
    in_len: usize,
    out_len: usize,
    /// `ffi::FL_LAYOUT_SEPARATE` / `ffi::FL_LAYOUT_ZONED` / `ffi::FL_LAYOUT_INTERLEAVED`: the layout that was kept
    pub layout: i32,
    /// GB/s of the probe stream per layout (0 = not measured)
    pub probe_gbps: [u32; ffi::FL_LAYOUT_COUNT],
    _own: PhantomData<T>,
}
impl<T> ColumnPair<T> {
    /// `in_len` / `out_len` in ELEMENTS of `T`; `layout` one of `ffi::FL_LAYOUT_*`
    pub fn new(in_len: usize, out_len: usize, layout: i32, stream: Stream) -> Self {
        let (mut i, mut o, mut h) = (core::ptr::null_mut::<c_void>(), core::ptr::null_mut::<c_void>(), core::ptr::null_mut::<c_void>());
        let (mut kept, mut gbps) = (-1i32, [0u32; ffi::FL_LAYOUT_COUNT]);
        ffi::check(
            unsafe {
                ffi::fl_column_pair_alloc(in_len * size_of::<T>(), 0, out_len * size_of::<T>(), layout, stream.0, &mut i,
                                          core::ptr::null_mut(), &mut o, &mut h, &mut kept, gbps.as_mut_ptr())
            },
            "fl_column_pair_alloc",
        );
        Self { handle: h, input: i as *mut T, output: o as *mut T, in_len, out_len, layout: kept, probe_gbps: gbps, _own: PhantomData }
    }
    /// the read side, to be filled by the caller (e.g. a `hipMemcpy` of the packed column)
    pub fn input_mut(&mut self) -> DeviceSliceMut<'_, T> { unsafe { DeviceSliceMut::from_raw_parts(self.input, self.in_len) } }
    pub fn input(&self) -> DeviceSlice<'_, T> { unsafe { DeviceSlice::from_raw_parts(self.input as *const T, self.in_len) } }
    pub fn output_mut(&mut self) -> DeviceSliceMut<'_, T> { unsafe { DeviceSliceMut::from_raw_parts(self.output, self.out_len) } }
    /// both sides at once (the two buffers never overlap): `let (packed, out) = pair.split(); codec.unpack(w, packed, out)`
    pub fn split(&mut self) -> (DeviceSlice<'_, T>, DeviceSliceMut<'_, T>) {
        unsafe { (DeviceSlice::from_raw_parts(self.input as *const T, self.in_len), DeviceSliceMut::from_raw_parts(self.output, self.out_len)) }
    }
}
impl<T> Drop for ColumnPair<T> {
    fn drop(&mut self) {
        unsafe { ffi::fl_column_pair_free(self.handle) };
    }
}
