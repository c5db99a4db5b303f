"""Host-side mirror of the reference's codec traits over the C ABI.

The reference (a Rust crate; no Rust toolchain in this image) exposes static
trait methods per element type:

    BitPacking::{pack::<W>, unpack::<W>, unpack_single::<W>, unchecked_pack,
                 unchecked_unpack, unchecked_unpack_single}   (bitpacking.rs:16-59)
    FoR::{for_pack::<W>, unfor_pack::<W>}                     (ffor.rs:4-18)
    Delta::{delta, undelta, undelta_pack::<W>}                (delta.rs:6-17)
    Transpose::{transpose, untranspose}                       (transpose.rs:4-7)

This module keeps those names and argument meanings.  The element type is taken
from the array dtype (u8/u16/u32/u64 by item size), the const generic W becomes
the runtime `width` argument (as in the reference's own `unchecked_*` methods),
and every method is batched over `n_blocks = len(input) / block_len` contiguous
1024-value blocks (n_blocks = 1 is exactly one trait call).

* torch CUDA tensors  -> device tier (asynchronous on the current HIP stream)
* numpy arrays        -> host tier   (staged through device memory, synchronous)

Errors follow the reference: width > T and index >= 1024 are panics there
(bitpacking.rs:93,126,152,197) and raise FastLanesError here.  There is no CPU
implementation behind any of these calls.
"""
import ctypes

import numpy as np

from . import _lib

_NP = {1: "u8", 2: "u16", 4: "u32", 8: "u64"}
_NP_DTYPE = {"u8": np.uint8, "u16": np.uint16, "u32": np.uint32, "u64": np.uint64}


class FastLanesError(RuntimeError):
    def __init__(self, status, where):
        lib = _lib.load()
        msg = lib.fl_status_string(status).decode()
        if status == 5:
            msg += f" (hipError_t {lib.fl_last_hip_error()})"
        super().__init__(f"{where}: {msg}")
        self.status = status


def packed_len(ty, width):
    """bitpacking.rs:77: elements per packed block = 1024 * W / T."""
    return 1024 * width // _lib.BITS[ty]


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _ty_of(x):
    if _is_torch(x):
        return _NP[x.element_size()]
    return _NP[np.asarray(x).dtype.itemsize]


def _check(rc, where):
    if rc != 0:
        raise FastLanesError(rc, where)


# FL_DEVERR_* bits of a device error flag -> the fl_status a synchronous call would have returned
_DEVERR = ((1, 1), (2, 2), (8, 6), (4, 4))     # width, index, bounds, align (most specific first)


def _check_flag(err, where):
    bits = int(err.item())
    for bit, status in _DEVERR:
        if bits & bit:
            raise FastLanesError(status, where)
    if bits:
        raise FastLanesError(5, where)


def _indices(index, like):
    """Column-global element indices for the device tier: a CUDA int64/uint64 tensor on `like`'s device (anything
    else is converted from the host).  A float tensor, or one of another device, would be reinterpreted / fault."""
    import torch
    if _is_torch(index):
        if index.dtype not in (torch.int64, torch.uint64):
            raise TypeError(f"indices must be an int64/uint64 tensor, got {index.dtype}")
        idx = _Arg(index.contiguous(), "u64")
        _same_tier(like, idx)
        return idx.x
    a = np.asarray(index)
    if a.dtype.kind not in "ui":
        raise TypeError(f"indices must be integers, got dtype {a.dtype}")
    return torch.as_tensor(a.astype(np.int64).reshape(-1), device=like.x.device)


class _Arg:
    """Uniform view of a numpy array or a torch CUDA tensor."""

    def __init__(self, x, ty=None):
        self.torch = _is_torch(x)
        if self.torch:
            if not x.is_cuda:
                raise TypeError("torch tensors must live on the GPU (use numpy arrays for the host tier)")
            if not x.is_contiguous():
                raise ValueError("tensor must be contiguous")
            self.x = x
            self.n = x.numel()
            self.ptr = x.data_ptr()
        else:
            a = np.asarray(x)
            if a.dtype.kind not in "ui":
                raise TypeError(f"expected an unsigned integer array, got dtype {a.dtype}")
            if ty is not None and a.dtype.itemsize * 8 != _lib.BITS[ty]:
                raise TypeError(f"expected a {ty} array, got dtype {a.dtype}")
            if not a.flags["C_CONTIGUOUS"]:
                raise ValueError("array must be C-contiguous (an output would otherwise be written to a copy)")
            a = a.view(_NP_DTYPE[_NP[a.dtype.itemsize]])   # signed ints are reinterpreted, never converted
            self.x = a
            self.n = a.size
            self.ptr = a.ctypes.data
        self.ty = ty or _ty_of(x)
        if self.torch and ty is not None and x.element_size() * 8 != _lib.BITS[ty]:
            raise TypeError(f"expected a {ty} tensor, got dtype {x.dtype}")


def _empty_like(arg, n, ty):
    if arg.torch:
        import torch
        return torch.empty(n, dtype=arg.x.dtype, device=arg.x.device)
    return np.empty(n, dtype=_NP_DTYPE[ty])


def _stream(arg):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(arg.x.device).cuda_stream)


def _same_tier(src, *others):
    """Every buffer of one call lives on the same tier (all numpy, or all CUDA tensors of ONE device):
    a host pointer handed to a device kernel -- or a tensor of another GPU -- would fault the process."""
    for o in others:
        if o is None:
            continue
        if o.torch != src.torch:
            raise TypeError("input, output and base/reference must all be numpy arrays (host tier) or all be "
                            "CUDA tensors (device tier)")
        if src.torch and o.x.device != src.x.device:
            raise ValueError(f"tensors live on different devices ({src.x.device} vs {o.x.device})")


def _out(src, output, ty, n_elems, what):
    """The caller's output buffer (checked: a short buffer would be overrun by the kernel, which sizes
    its stores from n_blocks) or a fresh one.  Mirrors the reference's length asserts
    (bitpacking.rs:78-80,111-113)."""
    if output is None:
        return _Arg(_empty_like(src, n_elems, ty), ty)
    out = _Arg(output, ty)
    if out.n != n_elems:
        raise ValueError(f"{what}: output holds {out.n} elements, expected {n_elems}")
    return out


def _consumer_out(src, output, dtype, n_elems, what):
    """Output tensor of a fused consumer (sums / mask): the caller's CUDA tensor of exactly n_elems elements of `dtype`'s width on
    the input's device, or a fresh one.  (Where a thin output stream lives relative to the packed input moves the rate by
    10-15 %: DESIGN.md 4, fastlanes_amd/placement.py.)"""
    import torch
    if output is None:
        return torch.empty(n_elems, dtype=dtype, device=src.x.device)
    if not (_is_torch(output) and output.is_cuda and output.is_contiguous()):
        raise TypeError(f"{what}: output must be a contiguous CUDA tensor")
    if output.device != src.x.device:
        raise ValueError(f"tensors live on different devices ({src.x.device} vs {output.device})")
    if output.element_size() != torch.empty(0, dtype=dtype).element_size() or output.numel() != n_elems:
        raise ValueError(f"{what}: output holds {output.numel()} x {output.element_size()}-byte elements, expected {n_elems} x "
                         f"{torch.empty(0, dtype=dtype).element_size()}")
    return output


def _run(method, ty, width, src, out, n_blocks, aux=None, aux_stride=None, scalar=None):
    """Dispatch to fl_<ty>_<method>[_host]."""
    lib = _lib.load()
    dev = src.torch
    _same_tier(src, out, aux)
    fn = getattr(lib, f"fl_{ty}_{method}" + ("" if dev else "_host"))
    args = []
    if width is not None:
        args.append(width)
    args.append(src.ptr)
    if method in ("for_pack", "unfor_pack"):
        if dev:
            args += [aux.ptr, aux_stride]
        else:
            args.append(scalar)
    elif aux is not None:
        args.append(aux.ptr)
    args += [out.ptr, n_blocks]
    if dev:
        import torch
        if src.x.device.index == torch.cuda.current_device():   # common case: skip the device context switch
            args.append(_stream(src))
            rc = fn(*args)
        else:
            with torch.cuda.device(src.x.device):
                args.append(_stream(src))
                rc = fn(*args)
    else:
        rc = fn(*args)
    _check(rc, f"fl_{ty}_{method}")
    return out.x


def _blocks(n, per_block, what):
    if per_block == 0:
        return None
    if n % per_block:
        raise ValueError(f"{what}: length {n} is not a multiple of {per_block}")
    return n // per_block


class BitPacking:
    """bitpacking.rs:16-59"""

    @staticmethod
    def pack(width, input, output=None):
        """BitPacking::pack::<W> / unchecked_pack (bitpacking.rs:19,30,65-96)."""
        src = _Arg(input)
        ty = src.ty
        if width > _lib.BITS[ty]:
            raise FastLanesError(1, f"fl_{ty}_pack")
        n = _blocks(src.n, 1024, "pack input")
        out = _out(src, output, ty, n * packed_len(ty, width), "pack")     # bitpacking.rs:78
        return _run("pack", ty, width, src, out, n)

    unchecked_pack = pack

    @staticmethod
    def unpack(width, input, output=None, n_blocks=None):
        """BitPacking::unpack::<W> / unchecked_unpack (bitpacking.rs:33,44,98-129).
        n_blocks is only needed for width == 0 (the packed input is empty)."""
        src = _Arg(input)
        ty = src.ty
        if width > _lib.BITS[ty]:
            raise FastLanesError(1, f"fl_{ty}_unpack")
        n = _blocks(src.n, packed_len(ty, width), "unpack input")
        if n is None:
            n = n_blocks if n_blocks is not None else (_Arg(output, ty).n // 1024 if output is not None else 0)
        out = _out(src, output, ty, n * 1024, "unpack")                    # bitpacking.rs:112
        return _run("unpack", ty, width, src, out, n)

    unchecked_unpack = unpack

    @staticmethod
    def unpack_single(width, packed, index, n_blocks=None):
        """BitPacking::unpack_single::<W> / unchecked_unpack_single (bitpacking.rs:47,58,132-200).
        `index` is an int (host tier: returns an int) or, on the device tier, a CUDA int64/uint64
        tensor of column-global element indices (block*1024 + i): returns a tensor of values."""
        lib = _lib.load()
        src = _Arg(packed)
        ty = src.ty
        if width > _lib.BITS[ty]:
            raise FastLanesError(1, f"fl_{ty}_unpack_single")
        pl = packed_len(ty, width)
        nb = n_blocks if n_blocks is not None else (src.n // pl if pl else 1)
        if not src.torch:
            val = _lib.CTYPE[ty](0)
            rc = getattr(lib, f"fl_{ty}_unpack_single_host")(width, src.ptr, nb, int(index), ctypes.byref(val))
            _check(rc, f"fl_{ty}_unpack_single")
            return val.value
        import torch
        idx = _indices(index, src)
        out = torch.empty(idx.numel(), dtype=src.x.dtype, device=src.x.device)
        err = torch.zeros(1, dtype=torch.int32, device=src.x.device)
        with torch.cuda.device(src.x.device):
            rc = getattr(lib, f"fl_{ty}_unpack_single")(width, src.ptr, nb, idx.data_ptr(), idx.numel(),
                                                        out.data_ptr(), err.data_ptr(), _stream(src))
        _check(rc, f"fl_{ty}_unpack_single")
        _check_flag(err, f"fl_{ty}_unpack_single")             # bitpacking.rs:152
        return out

    unchecked_unpack_single = unpack_single

    # ---- extensions -------------------------------------------------------
    @staticmethod
    def unpack_block_sums(width, packed, n_blocks=None, output=None):
        """sums[b] = sum(BitPacking.unpack(width, block b)) as wrapping uint64, without
        materialising the values.  Device tier only; returns a CUDA int64 tensor (bit pattern
        of the uint64 sums) -- `output` (n_blocks 8-byte elements) if given."""
        import torch
        src = _Arg(packed)
        ty = src.ty
        if width > _lib.BITS[ty]:
            raise FastLanesError(1, f"fl_{ty}_unpack_block_sums")
        n = _blocks(src.n, packed_len(ty, width), "unpack_block_sums input")
        if n is None:
            n = n_blocks or 0
        out = _consumer_out(src, output, torch.int64, n, "unpack_block_sums")
        with torch.cuda.device(src.x.device):
            _check(getattr(_lib.load(), f"fl_{ty}_unpack_block_sums")(width, src.ptr, n, out.data_ptr(), _stream(src)),
                   f"fl_{ty}_unpack_block_sums")
        return out

    CMP = {"==": 0, "!=": 1, "<": 2, "<=": 3, ">": 4, ">=": 5}

    @staticmethod
    def unpack_compare(width, packed, op, constant, n_blocks=None, output=None):
        """Selection mask straight from packed data: bit i of block b's 1024-bit mask =
        (BitPacking.unpack(width, block b)[i] <op> constant), op in '==','!=','<','<=','>','>='.
        Device tier only; returns a CUDA int32 tensor of 32 words per block (bit i of word i//32,
        LSB first) -- `output` (32 * n_blocks 4-byte elements) if given."""
        import torch
        src = _Arg(packed)
        ty = src.ty
        if width > _lib.BITS[ty]:
            raise FastLanesError(1, f"fl_{ty}_unpack_compare")
        n = _blocks(src.n, packed_len(ty, width), "unpack_compare input")
        if n is None:
            n = n_blocks or 0
        out = _consumer_out(src, output, torch.int32, n * 32, "unpack_compare")
        k = _lib.CTYPE[ty](int(constant) & ((1 << _lib.BITS[ty]) - 1))
        with torch.cuda.device(src.x.device):
            _check(getattr(_lib.load(), f"fl_{ty}_unpack_compare")(width, src.ptr, BitPacking.CMP[op], k, n,
                                                                   out.data_ptr(), _stream(src)),
                   f"fl_{ty}_unpack_compare")
        return out

    @staticmethod
    def block_min_max(values, output=None):
        """(mins, maxs) per 1024-value block of an unpacked column -- written into `output` = (mins, maxs), two CUDA tensors of
        n_blocks elements of the column's type, if given.  Device tier only."""
        import torch
        src = _Arg(values)
        ty = src.ty
        n = _blocks(src.n, 1024, "block_min_max input")
        mins, maxs = output if output is not None else (None, None)
        mins = _consumer_out(src, mins, src.x.dtype, n, "block_min_max mins")
        maxs = _consumer_out(src, maxs, src.x.dtype, n, "block_min_max maxs")
        with torch.cuda.device(src.x.device):
            _check(getattr(_lib.load(), f"fl_{ty}_block_min_max")(src.ptr, n, mins.data_ptr(), maxs.data_ptr(), _stream(src)),
                   f"fl_{ty}_block_min_max")
        return mins, maxs


class FoR:
    """ffor.rs:4-18.  `reference` is a scalar, or (device tier) a per-block CUDA tensor."""

    @staticmethod
    def _ref(src, ty, reference, n):
        if not src.torch:
            return None, None, _lib.CTYPE[ty](int(reference) & ((1 << _lib.BITS[ty]) - 1))
        import torch
        if _is_torch(reference):
            r = _Arg(reference.contiguous(), ty)
            _same_tier(src, r)
            if r.n not in (1, n):
                raise ValueError("references must hold 1 or n_blocks elements")
            return r, (0 if r.n == 1 else 1), None
        host = np.array([int(reference) & ((1 << _lib.BITS[ty]) - 1)], dtype=_NP_DTYPE[ty])
        r = torch.from_numpy(host.view(np.uint8)).to(src.x.device).view(src.x.dtype)
        return _Arg(r, ty), 0, None

    @staticmethod
    def for_pack(width, input, reference, output=None):
        """FoR::for_pack::<W> (ffor.rs:5-9,24-36)."""
        src = _Arg(input)
        ty = src.ty
        if width > _lib.BITS[ty]:
            raise FastLanesError(1, f"fl_{ty}_for_pack")
        n = _blocks(src.n, 1024, "for_pack input")
        out = _out(src, output, ty, n * packed_len(ty, width), "for_pack")
        aux, stride, scalar = FoR._ref(src, ty, reference, n)
        return _run("for_pack", ty, width, src, out, n, aux=aux, aux_stride=stride, scalar=scalar)

    @staticmethod
    def unfor_pack(width, input, reference, output=None, n_blocks=None):
        """FoR::unfor_pack::<W> (ffor.rs:11-17,38-50)."""
        src = _Arg(input)
        ty = src.ty
        if width > _lib.BITS[ty]:
            raise FastLanesError(1, f"fl_{ty}_unfor_pack")
        n = _blocks(src.n, packed_len(ty, width), "unfor_pack input")
        if n is None:   # width 0: the packed input is empty -- take the block count from n_blocks or the output
            n = n_blocks if n_blocks is not None else (_Arg(output, ty).n // 1024 if output is not None else 0)
        out = _out(src, output, ty, n * 1024, "unfor_pack")
        aux, stride, scalar = FoR._ref(src, ty, reference, n)
        return _run("unfor_pack", ty, width, src, out, n, aux=aux, aux_stride=stride, scalar=scalar)


class Delta:
    """delta.rs:6-17.  `base` holds LANES = 1024/T elements per block."""

    @staticmethod
    def _go(method, width, input, base, output, per_block, n_blocks=None):
        src = _Arg(input)
        ty = src.ty
        if width is not None and width > _lib.BITS[ty]:
            raise FastLanesError(1, f"fl_{ty}_{method}")
        n = _blocks(src.n, per_block(ty), f"{method} input")
        b = _Arg(base, ty)
        if n is None:
            n = n_blocks if n_blocks is not None else b.n // (1024 // _lib.BITS[ty])
        if b.n != n * (1024 // _lib.BITS[ty]):
            raise ValueError("base must hold LANES elements per block")
        out = _out(src, output, ty, n * 1024, method)
        return _run(method, ty, width, src, out, n, aux=b)

    @staticmethod
    def delta(input, base, output=None):
        """Delta::delta (delta.rs:7,24-33)."""
        return Delta._go("delta", None, input, base, output, lambda ty: 1024)

    @staticmethod
    def undelta(input, base, output=None):
        """Delta::undelta (delta.rs:9,36-45)."""
        return Delta._go("undelta", None, input, base, output, lambda ty: 1024)

    @staticmethod
    def undelta_pack(width, input, base, output=None):
        """Delta::undelta_pack::<W> (delta.rs:11-16,47-63); output is in transposed order."""
        return Delta._go("undelta_pack", width, input, base, output, lambda ty: packed_len(ty, width))


    # ---- extensions: the compositions delta.rs:88-100 performs, in one pass ----
    @staticmethod
    def undelta_pack_untranspose(width, input, base, output=None):
        """== Transpose.untranspose(Delta.undelta_pack(width, input, base)): decodes straight to
        the ORIGINAL element order.  Device tier only."""
        if not _is_torch(input):
            raise TypeError("undelta_pack_untranspose is a device-tier extension (pass CUDA tensors)")
        return Delta._go("undelta_pack_untranspose", width, input, base, output, lambda ty: packed_len(ty, width))

    @staticmethod
    def transpose_delta_pack(width, input, base, output=None):
        """== BitPacking.pack(width, Delta.delta(Transpose.transpose(input), base)): encodes straight
        from the ORIGINAL element order.  Device tier only."""
        if not _is_torch(input):
            raise TypeError("transpose_delta_pack is a device-tier extension (pass CUDA tensors)")
        src = _Arg(input)
        ty = src.ty
        if width > _lib.BITS[ty]:
            raise FastLanesError(1, f"fl_{ty}_transpose_delta_pack")
        n = _blocks(src.n, 1024, "transpose_delta_pack input")
        b = _Arg(base, ty)
        if b.n != n * (1024 // _lib.BITS[ty]):
            raise ValueError("base must hold LANES elements per block")
        out = _out(src, output, ty, n * packed_len(ty, width), "transpose_delta_pack")
        return _run("transpose_delta_pack", ty, width, src, out, n, aux=b)


class Transpose:
    """transpose.rs:4-7"""

    @staticmethod
    def _go(method, input, output):
        src = _Arg(input)
        ty = src.ty
        n = _blocks(src.n, 1024, f"{method} input")
        out = _out(src, output, ty, n * 1024, method)
        return _run(method, ty, None, src, out, n)

    @staticmethod
    def transpose(input, output=None):
        """Transpose::transpose (transpose.rs:5,11-15)."""
        return Transpose._go("transpose", input, output)

    @staticmethod
    def untranspose(input, output=None):
        """Transpose::untranspose (transpose.rs:6,17-22)."""
        return Transpose._go("untranspose", input, output)


def _check_widths_host(ty, widths):
    """Host widths -> contiguous uint8, refusing anything above T BEFORE the cast (300 must not wrap to 44)."""
    w = np.asarray(widths)
    if w.dtype.kind not in "ui":
        raise TypeError("widths must be an integer array")
    if w.size and (int(w.min()) < 0 or int(w.max()) > _lib.BITS[ty]):
        raise FastLanesError(1, "widths")                       # bitpacking.rs:93 unreachable!()
    return np.ascontiguousarray(w, dtype=np.uint8).reshape(-1)


def widths_to_offsets(ty, widths):
    """Device tier: (offsets, total) for a CUDA uint8 tensor of per-block widths -- offsets[b] = byte
    offset of block b in a back-to-back packed column (exclusive prefix sum of 128*W, bitpacking.rs:77),
    total = a 1-element CUDA int64 tensor holding the column's packed size.  No host round trip."""
    import torch
    w = _Arg(widths, "u8")
    if not w.torch:
        raise TypeError("widths_to_offsets is device tier (pass a CUDA uint8 tensor; sharding.packed_offsets is the host form)")
    dev = w.x.device
    offsets = torch.empty(w.n, dtype=torch.int64, device=dev)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _check(_lib.load().fl_widths_to_offsets(_lib.BITS[ty], w.ptr, w.n, offsets.data_ptr(), total.data_ptr(),
                                                err.data_ptr(), _stream(w)), "fl_widths_to_offsets")
    _check_flag(err, "fl_widths_to_offsets")                    # bitpacking.rs:93 unreachable!()
    return offsets, total


# the packed side comes first for the decoders, last for the encoders; FoR's references / Delta's bases sit between the two sides
_WIDTHS_DECODERS = ("unpack_widths", "unfor_pack_widths", "undelta_pack_widths", "undelta_pack_untranspose_widths")


def _widths_call(method, ty, widths, offsets, packed, unpacked, check, aux=()):
    """`aux`: the C arguments between the two sides -- (references pointer, stride) for FoR, (bases pointer,) for Delta."""
    import torch
    w = _Arg(widths, "u8")
    o = _Arg(offsets, "u64")
    _same_tier(packed, unpacked, w, o)
    if not packed.torch:
        raise TypeError(f"{method} is device tier: widths, offsets and data must be CUDA tensors")
    n = w.n
    if o.n != n:
        raise ValueError("offsets must hold one entry per block")
    if unpacked.n != n * 1024:
        raise ValueError(f"{method}: the unpacked column must hold 1024 elements per block")
    dev = packed.x.device
    err = torch.zeros(1, dtype=torch.int32, device=dev) if check else None
    pbytes = packed.n * (_lib.BITS[ty] // 8)     # the kernel skips (and flags) any block that does not lie inside these bytes
    # C ABI argument order is (widths, offsets, in, [aux], out, ..): packed, packed_bytes -> unpacked for the decoders, the reverse
    # for the encoders
    args = (packed.ptr, pbytes, *aux, unpacked.ptr) if method in _WIDTHS_DECODERS else (unpacked.ptr, *aux, packed.ptr, pbytes)
    with torch.cuda.device(dev):
        _check(getattr(_lib.load(), f"fl_{ty}_{method}")(w.ptr, o.ptr, *args, n,
                                                         err.data_ptr() if check else None, _stream(packed)),
               f"fl_{ty}_{method}")
    if check:
        _check_flag(err, f"fl_{ty}_{method}")                   # bitpacking.rs:93,126 unreachable!(); :78-80,111-113


def _block_references(src, ty, references, n):
    """FoR references of a mixed-width column: a CUDA tensor with one scalar per block, or one scalar for all (stride 0)."""
    r = _Arg(references, ty)
    _same_tier(src, r)
    if r.n not in (1, n):
        raise ValueError("references must hold one scalar per block (or exactly one, broadcast)")
    return r, (r.ptr, 0 if r.n == 1 and n != 1 else 1)


def _block_bases(src, ty, base, n):
    b = _Arg(base, ty)
    _same_tier(src, b)
    if b.n != n * (1024 // _lib.BITS[ty]):
        raise ValueError("base must hold LANES elements per block")
    return b, (b.ptr,)


def unpack_widths(widths, offsets, packed, output=None, check=True):
    """The reference's caller loop `for b: T::unchecked_unpack(widths[b], &packed[offsets[b]..], ..)`
    (bitpacking.rs:109-129) as ONE launch with everything device-resident: `widths` (CUDA uint8, one per
    block), `offsets` (CUDA int64/uint64 byte offsets into `packed`), `packed` (CUDA tensor of the element
    type).  `check=True` reads the device error flag back (one sync) and raises on a width > T, an
    offset that is not a multiple of 16, or a block that does not lie inside `packed` (the kernel skips such blocks
    either way); `check=False` stays asynchronous."""
    src = _Arg(packed)
    ty = src.ty
    n = _Arg(widths, "u8").n
    out = _out(src, output, ty, n * 1024, "unpack_widths")
    _widths_call("unpack_widths", ty, widths, offsets, src, out, check)
    return out.x


def pack_widths(widths, offsets, input, output, check=True):
    """`for b: T::unchecked_pack(widths[b], &input[b*1024..], &mut output[offsets[b]..])`
    (bitpacking.rs:76-96), device-resident.  `output` is the packed column (the caller sizes it from
    widths_to_offsets' total); bytes no block covers are left untouched."""
    src = _Arg(input)
    ty = src.ty
    out = _Arg(output, ty)
    _widths_call("pack_widths", ty, widths, offsets, out, src, check)
    return out.x


def unfor_pack_widths(widths, offsets, packed, references, output=None, check=True):
    """`for b: FoR::unfor_pack::<widths[b]>(&packed[offsets[b]..], references[b], ..)` (ffor.rs:38-50) -- unpack_widths with
    FoR's body; `references` is a CUDA tensor of one scalar per block (or a single one, broadcast)."""
    src = _Arg(packed)
    ty = src.ty
    n = _Arg(widths, "u8").n
    r, aux = _block_references(src, ty, references, n)
    out = _out(src, output, ty, n * 1024, "unfor_pack_widths")
    _widths_call("unfor_pack_widths", ty, widths, offsets, src, out, check, aux)
    return out.x


def for_pack_widths(widths, offsets, input, references, output, check=True):
    """`for b: FoR::for_pack::<widths[b]>(&input[b*1024..], references[b], &mut output[offsets[b]..])` (ffor.rs:24-36)."""
    src = _Arg(input)
    ty = src.ty
    out = _Arg(output, ty)
    r, aux = _block_references(src, ty, references, _Arg(widths, "u8").n)
    _widths_call("for_pack_widths", ty, widths, offsets, out, src, check, aux)
    return out.x


def for_widths(mins, maxs):
    """The encoder's step between block_min_max and for_pack_widths: widths[b] = bit length of maxs[b] - mins[b], the
    smallest W for which for_pack::<W>(block b, mins[b]) loses nothing (0 for a constant block).  CUDA tensors in, a CUDA
    uint8 tensor out; asynchronous.  (The reference selects no widths: this is its callers' arithmetic.)"""
    import torch
    lo = _Arg(mins)
    ty = lo.ty
    hi = _Arg(maxs, ty)
    _same_tier(lo, hi)
    if not lo.torch:
        raise TypeError("for_widths is device tier: pass CUDA tensors")
    if hi.n != lo.n:
        raise ValueError("mins and maxs must hold one entry per block")
    widths = torch.empty(lo.n, dtype=torch.uint8, device=lo.x.device)
    with torch.cuda.device(lo.x.device):
        _check(getattr(_lib.load(), f"fl_{ty}_for_widths")(lo.ptr, hi.ptr, lo.n, widths.data_ptr(), _stream(lo)), f"fl_{ty}_for_widths")
    return widths


def undelta_pack_widths(widths, offsets, packed, base, output=None, check=True, untranspose=False):
    """`for b: Delta::undelta_pack::<widths[b]>(&packed[offsets[b]..], &base[b], ..)` (delta.rs:47-63); the output is in
    transposed order like the reference's, or in ORIGINAL order with `untranspose=True` (the fused extension)."""
    src = _Arg(packed)
    ty = src.ty
    n = _Arg(widths, "u8").n
    b, aux = _block_bases(src, ty, base, n)
    method = "undelta_pack_untranspose_widths" if untranspose else "undelta_pack_widths"
    out = _out(src, output, ty, n * 1024, method)
    _widths_call(method, ty, widths, offsets, src, out, check, aux)
    return out.x


def transpose_delta_pack_widths(widths, offsets, input, base, output, check=True):
    """`for b: pack::<widths[b]>(delta(transpose(&input[b*1024..]), &base[b]))` into output[offsets[b]..] -- the fused encode
    extension (delta.rs:88-95 composed) over per-block widths."""
    src = _Arg(input)
    ty = src.ty
    out = _Arg(output, ty)
    b, aux = _block_bases(src, ty, base, _Arg(widths, "u8").n)
    _widths_call("transpose_delta_pack_widths", ty, widths, offsets, out, src, check, aux)
    return out.x


def unpack_single_widths(widths, offsets, packed, index):
    """`T::unchecked_unpack_single(widths[b], &packed[offsets[b]..], i)` (bitpacking.rs:58,181-200) batched over a mixed-width
    column: `index` is a CUDA int64/uint64 tensor of column-global element indices (block*1024 + i); returns the values.
    An index past the column raises like the reference's assert (bitpacking.rs:152), a width > T like its unreachable!()."""
    import torch
    src = _Arg(packed)
    ty = src.ty
    w = _Arg(widths, "u8")
    o = _Arg(offsets, "u64")
    _same_tier(src, w, o)
    if not src.torch:
        raise TypeError("unpack_single_widths is device tier: pass CUDA tensors")
    if o.n != w.n:
        raise ValueError("offsets must hold one entry per block")
    idx = _indices(index, src)
    out = torch.empty(idx.numel(), dtype=src.x.dtype, device=src.x.device)
    err = torch.zeros(1, dtype=torch.int32, device=src.x.device)
    with torch.cuda.device(src.x.device):
        _check(getattr(_lib.load(), f"fl_{ty}_unpack_single_widths")(w.ptr, o.ptr, src.ptr, src.n * (_lib.BITS[ty] // 8), w.n,
                                                                    idx.data_ptr(), idx.numel(), out.data_ptr(), err.data_ptr(),
                                                                    _stream(src)),
               f"fl_{ty}_unpack_single_widths")
    _check_flag(err, f"fl_{ty}_unpack_single_widths")
    return out


class Batch:
    """Many small arrays decoded / encoded in ONE launch (fl_<ty>_unpack_batch / _pack_batch): the shape of a columnar
    engine's chunks -- Vortex keeps 64 Ki values (64 blocks) per chunk and loops `unchecked_unpack` over each
    (bitpacking.rs:109-129), which is launch-bound as one call per chunk.  `packed[a]` / `unpacked[a]` are CUDA tensors of one
    element type on one device (array a: n_blocks[a] blocks of width widths[a]); the constructor uploads the four per-array
    device arrays (pointers, widths, block counts) once, unpack() / pack() are then one asynchronous launch each."""

    def __init__(self, packed, unpacked, widths, references=None, bases=None):
        """`references` (optional): one FoR reference per array -- unpack() / pack() then run unfor_pack::<W> / for_pack::<W>
        (ffor.rs:24-50) on every block of array a with references[a].
        `bases` (optional): one CUDA tensor per array holding its Delta bases, LANES elements per block (delta.rs:7) -- for
        undelta_pack() / transpose_delta_pack()."""
        import torch
        if not (len(packed) == len(unpacked) == len(widths)):
            raise ValueError("packed, unpacked and widths must have one entry per array")
        self.n = len(widths)
        args_p = [_Arg(t) for t in packed]
        args_u = [_Arg(t) for t in unpacked]
        if self.n == 0:
            raise ValueError("an empty batch has no element type")
        self.ty = args_u[0].ty
        T = _lib.BITS[self.ty]
        w = np.asarray(list(widths), dtype=np.int64)
        if (w < 0).any() or (w > T).any():
            raise FastLanesError(1, f"fl_{self.ty}_unpack_batch")      # bitpacking.rs:93,126 unreachable!()
        nb = []
        for a, (p, u) in enumerate(zip(args_p, args_u)):
            if p.ty != self.ty or u.ty != self.ty or not p.torch:
                raise TypeError("all arrays of a batch are CUDA tensors of one element type")
            _same_tier(args_u[0], p, u)
            if u.n % 1024:
                raise ValueError(f"array {a}: the unpacked array must hold 1024 elements per block")
            if p.n != (u.n // 1024) * packed_len(self.ty, int(w[a])):
                raise ValueError(f"array {a}: packed length {p.n} does not match {u.n // 1024} blocks of width {int(w[a])}")
            # the kernel can only SKIP a misaligned array and raise FL_DEVERR_ALIGN (the pointers reach it through HBM), which a
            # caller running with check=False would never see: here, where the pointers are still on the host, it is an error like
            # at every other entry point (FL_ERR_ALIGN).  A width-0 array has no packed bytes: any pointer, or none, will do.
            if u.ptr % 16 or (int(w[a]) != 0 and u.n and p.ptr % 16):
                raise FastLanesError(4, f"fl_{self.ty}_unpack_batch (array {a}: device pointers must be 16-byte aligned)")
            nb.append(u.n // 1024)
        self.device = args_u[0].x.device
        self._keep = (list(packed), list(unpacked))                     # the pointer arrays below refer to these
        up = lambda a, dt: torch.from_numpy(np.asarray(a, dtype=dt)).to(self.device)
        self.d_packed = up([a.ptr for a in args_p], np.int64)
        self.d_unpacked = up([a.ptr for a in args_u], np.int64)
        self.d_widths = up(w, np.uint8)
        self.d_n_blocks = up(nb, np.int32)
        self.max_blocks = max(nb)
        self.d_refs = None
        if references is not None:
            r = [int(x) & ((1 << T) - 1) for x in references]          # python ints: a u64 reference may exceed int64
            if len(r) != self.n:
                raise ValueError("references must hold one entry per array")
            self.d_refs = up(np.array(r, dtype=np.uint64).astype(_NP_DTYPE[self.ty]).view(np.uint8), np.uint8)
        self.unpacked = list(unpacked)
        self.packed = list(packed)
        self.d_bases = None
        if bases is not None:
            if len(bases) != self.n:
                raise ValueError("bases must hold one tensor per array")
            args_b = [_Arg(t, self.ty) for t in bases]
            for a, bb in enumerate(args_b):
                _same_tier(args_u[0], bb)
                if bb.n != nb[a] * (1024 // T):
                    raise ValueError(f"array {a}: bases must hold LANES elements per block")
                if nb[a] and bb.ptr % 16:
                    raise FastLanesError(4, f"fl_{self.ty}_undelta_pack_batch (array {a}: device pointers must be 16-byte aligned)")
            self._keep += (list(bases),)
            self.d_bases = up([a.ptr for a in args_b], np.int64)

    def _run_delta(self, method, extra, check):
        import torch
        if self.d_bases is None:
            raise ValueError("this batch was built without bases")
        err = torch.zeros(1, dtype=torch.int32, device=self.device) if check else None
        first, last = (self.d_packed, self.d_unpacked) if method == "undelta_pack_batch" else (self.d_unpacked, self.d_packed)
        with torch.cuda.device(self.device):
            st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _check(getattr(_lib.load(), f"fl_{self.ty}_{method}")(first.data_ptr(), self.d_bases.data_ptr(), last.data_ptr(), self.d_widths.data_ptr(),
                                                                  self.d_n_blocks.data_ptr(), self.n, self.max_blocks, *extra,
                                                                  err.data_ptr() if check else None, st), f"fl_{self.ty}_{method}")
        if check:
            _check_flag(err, f"fl_{self.ty}_{method}")

    def undelta_pack(self, untranspose=False, check=False):
        """Delta::undelta_pack::<widths[a]> (delta.rs:47-63) on every block of every array with its bases; the output is in
        transposed order like the reference's, or in ORIGINAL order with `untranspose=True` (the fused extension)."""
        self._run_delta("undelta_pack_batch", (1 if untranspose else 0,), check)
        return self.unpacked

    def transpose_delta_pack(self, check=False):
        """The fused encode pack::<widths[a]>(delta(transpose(unpacked[a]), bases[a])) for every array."""
        self._run_delta("transpose_delta_pack_batch", (), check)
        return self.packed

    def _run(self, method, first, second, check):
        import torch
        err = torch.zeros(1, dtype=torch.int32, device=self.device) if check else None
        if self.d_refs is not None:
            method = {"unpack_batch": "unfor_pack_batch", "pack_batch": "for_pack_batch"}[method]
        with torch.cuda.device(self.device):
            st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            args = [first.data_ptr(), second.data_ptr(), self.d_widths.data_ptr()]
            if self.d_refs is not None:
                args.append(self.d_refs.data_ptr())
            args += [self.d_n_blocks.data_ptr(), self.n, self.max_blocks, err.data_ptr() if check else None, st]
            _check(getattr(_lib.load(), f"fl_{self.ty}_{method}")(*args), f"fl_{self.ty}_{method}")
        if check:
            _check_flag(err, f"fl_{self.ty}_{method}")

    def unpack(self, check=False):
        """packed[a] -> unpacked[a] for every array; returns the list of unpacked tensors."""
        self._run("unpack_batch", self.d_packed, self.d_unpacked, check)
        return self.unpacked

    def pack(self, check=False):
        """unpacked[a] -> packed[a] for every array; returns the list of packed tensors."""
        self._run("pack_batch", self.d_unpacked, self.d_packed, check)
        return self.packed


class MixedWidthPlan:
    """A column whose blocks each have their own width (BASELINE.json config 5): the
    reference's caller loop `for b: T::unchecked_unpack(widths[b], ..)` (bitpacking.rs:109-129)
    as one call.  Built from the host `widths` array: uploads them and prefix-sums the packed byte
    offsets (128*widths[b], back to back) ON THE DEVICE; `widths` / `offsets` are then device arrays
    (see unpack_widths / pack_widths for callers whose widths already live in HBM)."""

    def __init__(self, ty, widths, device=None):
        import torch
        self.ty = ty
        w = _check_widths_host(ty, widths)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("a MixedWidthPlan lives on a GPU")
        if self.device.index is None:        # torch.device('cuda') != torch.device('cuda:0'): pin the index now
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._plan = ctypes.c_void_p()
        lib = _lib.load()
        with torch.cuda.device(self.device):
            _check(lib.fl_mixed_plan_create(_lib.BITS[ty], w.ctypes.data, w.size, ctypes.byref(self._plan)),
                   "fl_mixed_plan_create")
        self.n_blocks = int(lib.fl_mixed_plan_n_blocks(self._plan))
        self.packed_bytes = int(lib.fl_mixed_plan_packed_bytes(self._plan))

    def close(self):
        if getattr(self, "_plan", None):
            _lib.load().fl_mixed_plan_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: the library may already be gone
            pass

    def _torch_dtype(self):
        import torch
        return {"u8": torch.uint8, "u16": torch.uint16, "u32": torch.uint32, "u64": torch.uint64}[self.ty]

    def _on_device(self, *args):
        for a in args:
            if not a.torch or a.x.device != self.device:
                raise ValueError(f"this plan lives on {self.device}: pass CUDA tensors of that device")

    def unpack(self, packed, output=None):
        import torch
        src = _Arg(packed, self.ty)
        esz = _lib.BITS[self.ty] // 8
        if src.n * esz != self.packed_bytes:
            raise ValueError("packed column has the wrong size for this plan")
        out = _Arg(output, self.ty) if output is not None else _Arg(
            torch.empty(self.n_blocks * 1024, dtype=self._torch_dtype(), device=self.device), self.ty)
        if out.n != self.n_blocks * 1024:
            raise ValueError("Output buffer must be of size 1024 per block")
        self._on_device(src, out)
        with torch.cuda.device(self.device):
            _check(getattr(_lib.load(), f"fl_{self.ty}_unpack_mixed")(self._plan, src.ptr, out.ptr, _stream(out)),
                   f"fl_{self.ty}_unpack_mixed")
        return out.x

    def pack(self, input, output=None):
        import torch
        src = _Arg(input, self.ty)
        esz = _lib.BITS[self.ty] // 8
        if src.n != self.n_blocks * 1024:
            raise ValueError("Input buffer must be of size 1024 per block")
        out = _Arg(output, self.ty) if output is not None else _Arg(
            torch.empty(self.packed_bytes // esz, dtype=self._torch_dtype(), device=self.device), self.ty)
        if out.n * esz != self.packed_bytes:
            raise ValueError("packed column has the wrong size for this plan")
        self._on_device(src, out)
        with torch.cuda.device(self.device):
            _check(getattr(_lib.load(), f"fl_{self.ty}_pack_mixed")(self._plan, src.ptr, out.ptr, _stream(src)),
                   f"fl_{self.ty}_pack_mixed")
        return out.x
