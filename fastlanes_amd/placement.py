"""Zone-aware placement of a column's input and output in HBM.

Measured on MI355X (tools/abplacement*.cpp, tools/exp_zones.py; profiles/abplacement_r03.txt, profiles/abzones_r03.txt): the device
memory behaves as 64-GiB ZONES, and reads and writes want opposite things from them:

  * two concurrent WRITE streams inside one zone sustain 5.6 TB/s together, in two different zones 7.05 TB/s (+26 %);
  * two concurrent READ streams inside one zone sustain 6.8 TB/s together, in two different zones 6.35 TB/s (-7 %).

The codec kernels (all of whose global accesses are streaming reads or writes) follow: unpack u32 W=7 runs at ~6.2 TB/s when its
packed input and its unpacked output share a zone, ~6.75 when they lie in different zones and 6.85-6.9 when the OUTPUT is split
half and half over two zones -- the whole "which allocation did the buffers land in" spread of the bench numbers (0.75-0.86 of the
HBM peak) -- while a read-dominated pack loses when its INPUT is split.  Whether two separate allocations share a zone is the
driver's choice; inside ONE allocation it is the caller's: in every process measured, the zone boundaries of a large allocation
lay at multiples of 64 GiB from its start.

`column_pair` carves both buffers from one allocation: the INPUT (and aux) at offset 0 -- inside one zone as long as it is shorter
than 64 GiB --, the OUTPUT centred on the first 64-GiB multiple that leaves room for the input in front of it, so that the writes
are split over two zones.  If the allocation does not start on the zone grid after all, the worst case is still "input and output
in different zones" (the span exceeds one zone).  It is an allocation helper, nothing else: the codec entry points take any 16-byte
aligned device pointers.  The price is the unused memory between the two buffers (a column store would keep other columns there).
"""

ZONE_BYTES = 64 << 30
_ALIGN = 256


def _layout(in_bytes, out_bytes, aux_bytes):
    """byte offsets (input, aux, output) and the slab size: input (+ aux) at 0, output centred on the first multiple of ZONE_BYTES
    that leaves room for the input in front of it"""
    pad = lambda b: (b + _ALIGN - 1) & ~(_ALIGN - 1)
    in_end = pad(in_bytes) + pad(aux_bytes)
    half = pad(out_bytes) // 2
    k = 1
    while k * ZONE_BYTES - half < in_end:
        k += 1
        if k > 8:
            raise ValueError("buffers too large for a zone-aware layout")
    out_off = (k * ZONE_BYTES - half) & ~(_ALIGN - 1)
    return 0, pad(in_bytes), out_off, out_off + pad(out_bytes)


def column_pair(in_bytes, out_bytes, device, aux_bytes=0):
    """(slab, input, aux, output): uint8 views of one torch allocation (layout: module docstring).  `aux` (aux_bytes, e.g. Delta's
    bases) sits right behind the input.  Keep `slab` alive as long as the views."""
    import torch
    i_off, a_off, o_off, total = _layout(in_bytes, out_bytes, aux_bytes)
    slab = torch.empty(total, dtype=torch.uint8, device=device)
    return slab, slab[i_off:i_off + in_bytes], slab[a_off:a_off + aux_bytes], slab[o_off:o_off + out_bytes]


def fits(in_bytes, out_bytes, device, aux_bytes=0, reserve=2 << 30):
    """Is there room for column_pair() on `device` right now?"""
    import torch
    try:
        total = _layout(in_bytes, out_bytes, aux_bytes)[3]
    except ValueError:
        return False
    free, _ = torch.cuda.mem_get_info(device)
    return total + reserve <= free
