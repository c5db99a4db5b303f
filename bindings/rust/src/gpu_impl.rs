//! `impl BitPacking / FoR / Delta / Transpose` for u8/u16/u32/u64 over `libfastlanes_amd.so`,
//! replacing `impl_packing!` (bitpacking.rs:61-237), `impl_for!` (ffor.rs:20-59), `impl_delta!`
//! (delta.rs:19-73) and the blanket `Transpose` impl (transpose.rs:9-23) when the `gpu` feature is
//! on.  The const generic `W` becomes the runtime `width` argument, exactly as in the reference's own
//! `unchecked_*` methods.  Uncompiled: see ../README.md.
use core::mem::size_of;

use crate::ffi;
use crate::{BitPackWidth, BitPacking, Delta, FastLanes, FoR, SupportedBitPackWidth, Transpose};

macro_rules! impl_gpu {
    ($T:ty, $pack:ident, $unpack:ident, $single:ident, $for_pack:ident, $unfor_pack:ident,
     $delta:ident, $undelta:ident, $undelta_pack:ident, $transpose:ident, $untranspose:ident) => {
        impl BitPacking for $T {
            fn pack<const W: usize>(input: &[Self; 1024], output: &mut [Self; 1024 * W / Self::T])
            where BitPackWidth<W>: SupportedBitPackWidth<Self> {
                ffi::check(unsafe { ffi::$pack(W as u32, input.as_ptr(), output.as_mut_ptr(), 1) }, "pack");
            }

            unsafe fn unchecked_pack(width: usize, input: &[Self], output: &mut [Self]) {
                debug_assert_eq!(output.len(), 128 * width / size_of::<Self>(), "Output buffer must be of size 1024 * W / T");
                debug_assert_eq!(input.len(), 1024, "Input buffer must be of size 1024");
                ffi::check(ffi::$pack(width as u32, input.as_ptr(), output.as_mut_ptr(), 1), "unchecked_pack");
            }

            fn unpack<const W: usize>(input: &[Self; 1024 * W / Self::T], output: &mut [Self; 1024])
            where BitPackWidth<W>: SupportedBitPackWidth<Self> {
                ffi::check(unsafe { ffi::$unpack(W as u32, input.as_ptr(), output.as_mut_ptr(), 1) }, "unpack");
            }

            unsafe fn unchecked_unpack(width: usize, input: &[Self], output: &mut [Self]) {
                debug_assert_eq!(input.len(), 128 * width / size_of::<Self>(), "Input buffer must be of size 1024 * W / T");
                debug_assert_eq!(output.len(), 1024, "Output buffer must be of size 1024");
                ffi::check(ffi::$unpack(width as u32, input.as_ptr(), output.as_mut_ptr(), 1), "unchecked_unpack");
            }

            fn unpack_single<const W: usize>(packed: &[Self; 1024 * W / Self::T], index: usize) -> Self
            where BitPackWidth<W>: SupportedBitPackWidth<Self> {
                let mut value: Self = 0;
                ffi::check(unsafe { ffi::$single(W as u32, packed.as_ptr(), 1, index as u64, &mut value) }, "unpack_single");
                value
            }

            unsafe fn unchecked_unpack_single(width: usize, packed: &[Self], index: usize) -> Self {
                debug_assert_eq!(packed.len(), 128 * width / size_of::<Self>());
                let mut value: Self = 0;
                ffi::check(ffi::$single(width as u32, packed.as_ptr(), 1, index as u64, &mut value), "unchecked_unpack_single");
                value
            }
        }

        impl FoR for $T {
            fn for_pack<const W: usize>(input: &[Self; 1024], reference: Self, output: &mut [Self; 1024 * W / Self::T])
            where BitPackWidth<W>: SupportedBitPackWidth<Self> {
                ffi::check(unsafe { ffi::$for_pack(W as u32, input.as_ptr(), reference, output.as_mut_ptr(), 1) }, "for_pack");
            }

            fn unfor_pack<const W: usize>(input: &[Self; 1024 * W / Self::T], reference: Self, output: &mut [Self; 1024])
            where BitPackWidth<W>: SupportedBitPackWidth<Self> {
                ffi::check(unsafe { ffi::$unfor_pack(W as u32, input.as_ptr(), reference, output.as_mut_ptr(), 1) }, "unfor_pack");
            }
        }

        impl Delta for $T {
            fn delta(input: &[Self; 1024], base: &[Self; Self::LANES], output: &mut [Self; 1024]) {
                ffi::check(unsafe { ffi::$delta(input.as_ptr(), base.as_ptr(), output.as_mut_ptr(), 1) }, "delta");
            }

            fn undelta(input: &[Self; 1024], base: &[Self; Self::LANES], output: &mut [Self; 1024]) {
                ffi::check(unsafe { ffi::$undelta(input.as_ptr(), base.as_ptr(), output.as_mut_ptr(), 1) }, "undelta");
            }

            fn undelta_pack<const W: usize>(input: &[Self; 1024 * W / Self::T], base: &[Self; Self::LANES], output: &mut [Self; 1024])
            where BitPackWidth<W>: SupportedBitPackWidth<Self> {
                ffi::check(unsafe { ffi::$undelta_pack(W as u32, input.as_ptr(), base.as_ptr(), output.as_mut_ptr(), 1) },
                           "undelta_pack");
            }
        }

        impl Transpose for $T {
            fn transpose(input: &[Self; 1024], output: &mut [Self; 1024]) {
                ffi::check(unsafe { ffi::$transpose(input.as_ptr(), output.as_mut_ptr(), 1) }, "transpose");
            }

            fn untranspose(input: &[Self; 1024], output: &mut [Self; 1024]) {
                ffi::check(unsafe { ffi::$untranspose(input.as_ptr(), output.as_mut_ptr(), 1) }, "untranspose");
            }
        }
    };
}

impl_gpu!(u8, fl_u8_pack_host, fl_u8_unpack_host, fl_u8_unpack_single_host, fl_u8_for_pack_host, fl_u8_unfor_pack_host,
          fl_u8_delta_host, fl_u8_undelta_host, fl_u8_undelta_pack_host, fl_u8_transpose_host, fl_u8_untranspose_host);
impl_gpu!(u16, fl_u16_pack_host, fl_u16_unpack_host, fl_u16_unpack_single_host, fl_u16_for_pack_host, fl_u16_unfor_pack_host,
          fl_u16_delta_host, fl_u16_undelta_host, fl_u16_undelta_pack_host, fl_u16_transpose_host, fl_u16_untranspose_host);
impl_gpu!(u32, fl_u32_pack_host, fl_u32_unpack_host, fl_u32_unpack_single_host, fl_u32_for_pack_host, fl_u32_unfor_pack_host,
          fl_u32_delta_host, fl_u32_undelta_host, fl_u32_undelta_pack_host, fl_u32_transpose_host, fl_u32_untranspose_host);
impl_gpu!(u64, fl_u64_pack_host, fl_u64_unpack_host, fl_u64_unpack_single_host, fl_u64_for_pack_host, fl_u64_unfor_pack_host,
          fl_u64_delta_host, fl_u64_undelta_host, fl_u64_undelta_pack_host, fl_u64_transpose_host, fl_u64_untranspose_host);
