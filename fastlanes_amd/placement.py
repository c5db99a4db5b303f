"""Zone-aware placement of a column's input and output in HBM.

Measured on MI355X (tools/abplacement2.cpp, abplacement3.cpp, tools/exp_zones.py; profiles/abplacement_r03.txt,
profiles/abzones_r03.txt): the device memory behaves as 64-GiB ZONES.  Two concurrent write streams inside one zone sustain
5.6 TB/s together, in two different zones 7.05 TB/s; the codec kernels (every global access of theirs is a streaming read or
write) run at 6.2 TB/s when a column's packed input and its unpacked output share a zone and at 6.75-6.9 TB/s when they do
not -- the whole "which allocation did the buffers land in" spread of the bench numbers (0.78-0.86 of the HBM peak).  Whether
two separate allocations share a zone is the driver's choice; inside ONE allocation it is the caller's: byte offsets 64 GiB
apart are in different zones.

`column_pair` returns an input and an output buffer carved from one allocation, the output starting exactly one zone after the
input.  It is an allocation helper, nothing else: the codec entry points take any 16-byte aligned device pointers.  The price is
the memory between the end of the input and the start of the output (a real column store would put other columns there).
"""

ZONE_BYTES = 64 << 30


def column_pair(in_bytes, out_bytes, device, aux_bytes=0):
    """(slab, input, aux, output): uint8 views of one torch allocation; `input` (in_bytes) and `aux` (aux_bytes, e.g. Delta's bases)
    at the start, `output` (out_bytes) exactly ZONE_BYTES after the input's first byte.  Keep `slab` alive as long as the views."""
    import torch
    if in_bytes + aux_bytes + 256 > ZONE_BYTES:
        raise ValueError("input + aux must fit below the zone boundary")
    slab = torch.empty(ZONE_BYTES + out_bytes, dtype=torch.uint8, device=device)
    aux_off = (in_bytes + 255) & ~255
    return slab, slab[:in_bytes], slab[aux_off:aux_off + aux_bytes], slab[ZONE_BYTES:ZONE_BYTES + out_bytes]


def fits(in_bytes, out_bytes, device, aux_bytes=0, reserve=2 << 30):
    """Is there room for column_pair() on `device` right now?"""
    import torch
    free, _ = torch.cuda.mem_get_info(device)
    return in_bytes + aux_bytes + 256 <= ZONE_BYTES and ZONE_BYTES + out_bytes + reserve <= free
