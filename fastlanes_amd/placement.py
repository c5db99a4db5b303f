"""Layout helpers for MEASURING where a column's buffers live in HBM -- not a policy, and never needed to use the codec (every
entry point takes any 16-byte aligned device pointers).

What round 3 measured on MI355X (tools/abplacement*.cpp, tools/exp_*.py; profiles/abplacement_r03.txt, abzones_r03.txt,
exp_region_map_r03.txt): every 8-GiB granule of an allocation behaves as one of three classes of memory, in an order the driver
chooses (runs of 32-64 GiB at the start of a fresh allocation, shorter and differently ordered further in and in every process);
two concurrent WRITE streams sustain 5.6 TB/s together inside one stretch and 7.05 TB/s in two, two concurrent READ streams do the
opposite (6.8 / 6.35); a thin write stream next to a bulk read stream of the same class costs the consumers up to 14 %.  So the same
kernel moves by a few per cent with the allocation its buffers happen to lie in -- and nothing in an address tells.

What the repository does with that since round 4:
  * the KERNELS use the read / write asymmetry themselves (the tile-map window, fl_kernels.hpp: xcd_tile);
  * bench.py --placement auto treats the layout as a measurement: it allocates both layouts it knows -- one allocation per buffer,
    and `column_pair` below -- times a few launches on each BEFORE its timed region and keeps the faster buffers.  Round 3 shipped
    `column_pair` as the default and it cost one workload 9 %;
  * tools/sweep.py carves every row from one slab and places its consumers' thin outputs with `consumer_pair` (classes by a probe
    kernel), so that rows of one sweep are comparable.

`column_pair` carves both buffers from one allocation: the INPUT (and aux) at offset 0, the OUTPUT centred on the first 64-GiB
multiple that leaves room for the input in front of it -- in a fresh allocation that is where the first class boundary usually
lies, so the kernel's concurrent writes fall into two classes.  The price is the unused memory between the two buffers.
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


# ---------------------------------------------------------------------------------------------------------------------------
# A thin WRITE stream next to a bulk READ stream (a selection mask, per-block sums: 1/8 .. 1/500 of the bytes) is the one case where
# the layout above is not enough: the same unpack_compare runs at 7.0 TB/s or at 6.05 TB/s depending only on whether the mask lies
# in memory of the same CLASS as the packed input (profiles/exp_region_map_r03.txt: every 8-GiB granule of an allocation belongs to
# one of three classes -- most likely the three ranks of the HBM stacks --, in an order that differs from process to process).
# Nothing in an address tells the class, but a 1-ms kernel does: consumer_pair() times a small unpack_compare with its mask in
# every candidate granule and puts the output where it was fastest.
# ---------------------------------------------------------------------------------------------------------------------------
GRANULE_BYTES = 8 << 30
_PROBE_BLOCKS = 2_000_000          # u32 W=20: 5.1 GB read (inside one granule), 0.26 GB mask
_PROBE_WIDTH = 20


def _probe_rate(lib, slab, in_granule, mask_granule, reps=3):
    """GB/s of the probe kernel reading from the start of one granule of `slab` and writing its mask into the last GiB of another"""
    import ctypes
    import torch
    n, w = _PROBE_BLOCKS, _PROBE_WIDTH
    ib, ob = n * 128 * w, n * 128
    io = in_granule * GRANULE_BYTES
    oo = mask_granule * GRANULE_BYTES + GRANULE_BYTES - (1 << 30)
    src, dst = slab[io:io + ib], slab[oo:oo + ob]
    k = ctypes.c_uint32((1 << w) // 2)
    # on torch's CURRENT stream -- the one the torch events below are recorded on (the NULL stream would not be bracketed by
    # events of a non-blocking side stream)
    st = ctypes.c_void_p(torch.cuda.current_stream(slab.device).cuda_stream)
    def run():
        # under the whole-column tile map the classes were characterised with (policy window 31: fastlanes_amd_internal.h); a
        # windowed read stream interferes less with the thin write stream -- the point of the window -- and blunts the probe
        saved = lib.fl_internal_get_kernel_policy()
        lib.fl_internal_set_kernel_policy(31 << 25)
        try:
            return lib.fl_u32_unpack_compare(w, src.data_ptr(), 2, k, n, dst.data_ptr(), st)
        finally:
            lib.fl_internal_set_kernel_policy(saved)
    ms = []
    for i in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        if run() != 0:
            raise RuntimeError("placement probe: fl_u32_unpack_compare failed")
        b.record()
        b.synchronize()
        if i:
            ms.append(a.elapsed_time(b))
    return (ib + ob) / sorted(ms)[len(ms) // 2] / 1e6


def granule_classes(slab, lib=None):
    """Class (0, 1, 2, or None where the probe finds no clean answer) of every 8-GiB granule of `slab` (a uint8 CUDA tensor of
    a whole number of granules), by measurement: the probe kernel reads the start of a representative granule and writes its mask
    into the last GiB of every other one; the slow ones are of the representative's class (profiles/exp_region_map_r03.txt: 7.0 vs
    6.05 TB/s, nothing in between).  Overwrites parts of the slab.  Returns (classes, {class: {granule: GB/s against its
    representative}})."""
    import ctypes
    import torch
    from . import _lib
    lib = lib or _lib.load()
    n = slab.numel() // GRANULE_BYTES
    cls, rates, threshold = [None] * n, {}, None
    with torch.cuda.device(slab.device):
        for c in range(3):
            rep = next((g for g in range(n) if cls[g] is None), None)
            if rep is None:
                break
            cls[rep] = c
            others = [g for g in range(n) if cls[g] is None]
            if not others:
                break
            # full-entropy probe input (constant data would raise the clocks)
            st = ctypes.c_void_p(torch.cuda.current_stream(slab.device).cuda_stream)
            if lib.fl_fill_random(slab[rep * GRANULE_BYTES:].data_ptr(), _PROBE_BLOCKS * 128 * _PROBE_WIDTH, 17 + rep, st) != 0:
                raise RuntimeError("placement probe: fl_fill_random failed")
            rates[c] = {g: _probe_rate(lib, slab, rep, g) for g in others}
            hi, lo = max(rates[c].values()), min(rates[c].values())
            if threshold is None:
                if hi - lo <= 0.05 * hi:                   # one level only: no way to tell "all of my class" from "none of it"
                    break
                threshold = (hi + lo) / 2
            for g, r in rates[c].items():
                if r < threshold:
                    cls[g] = c
    return cls, rates


def choose_granules(cls, rates, k):
    """(first input granule, output granule, input_one_class) from granule_classes()' answer, for an input of k granules: the
    first run of k granules of one class and a granule of another class; without such a run, offset 0 and the granule that probed
    fastest against granule 0."""
    n = len(cls)
    start = next((i for i in range(n - k + 1) if cls[i] is not None and all(cls[j] == cls[i] for j in range(i, i + k))), None)
    one_class = start is not None
    start = start or 0
    outside = [g for g in range(n) if not start <= g < start + k]
    clean = [g for g in outside if cls[g] is not None and cls[g] != cls[start]] if one_class else []
    # any granule of another class will do; prefer one whose rate against the input's class was measured, highest first
    known = rates.get(cls[start] if clean else 0, {})
    best = max(clean or outside, key=lambda g: known.get(g, 0.0))
    return start, best, one_class


def consumer_pair(in_bytes, out_bytes, device, slab_bytes=128 << 30):
    """(slab, input, output, info) for a read-dominated consumer, both carved from one torch allocation of at least `slab_bytes`:
    the input in the first run of 8-GiB granules that are all of ONE class of memory, the output at the start of a granule of
    another class -- classes by measurement (granule_classes).  If no such run exists (fragmented memory) the input goes to offset
    0 and the output to the granule that probed fastest against granule 0.  `info` = {"classes": "AABB.C..", "input_granule",
    "output_granule", "input_one_class"}.  The probe overwrites parts of the slab: call this BEFORE filling the buffers.  Falls
    back to column_pair()'s layout if the output does not fit one granule."""
    import torch
    from . import _lib
    pad = lambda b: (b + _ALIGN - 1) & ~(_ALIGN - 1)
    k = max(1, (pad(in_bytes) + GRANULE_BYTES - 1) // GRANULE_BYTES)
    total = max(int(slab_bytes), (k + 2) * GRANULE_BYTES)
    total = (total + GRANULE_BYTES - 1) // GRANULE_BYTES * GRANULE_BYTES
    if pad(out_bytes) > GRANULE_BYTES - (1 << 30):
        slab, src, _, dst = column_pair(in_bytes, out_bytes, device)
        return slab, src, dst, {"classes": "", "input_granule": 0, "output_granule": None, "input_one_class": None}
    with torch.cuda.device(device):
        slab = torch.empty(total, dtype=torch.uint8, device=device)
    cls, rates = granule_classes(slab, _lib.load())
    start, best, one_class = choose_granules(cls, rates, k)
    i_off, o_off = start * GRANULE_BYTES, best * GRANULE_BYTES
    return slab, slab[i_off:i_off + in_bytes], slab[o_off:o_off + out_bytes], {
        "classes": "".join("." if c is None else "ABC"[c] for c in cls), "input_granule": start, "output_granule": best,
        "input_one_class": one_class}


# ---------------------------------------------------------------------------------------------------------------------------
# The C ABI's optional allocation helper (include/fastlanes_amd.h: fl_column_pair_alloc / _free), mirrored: what bench.py's
# --placement auto uses since round 5, so that the figure it prints is one the header alone reproduces.
# ---------------------------------------------------------------------------------------------------------------------------
LAYOUTS = {"separate": 0, "zoned": 1, "auto": 2, "interleaved": 3}
LAYOUT_NAMES = {0: "separate", 1: "zoned", 3: "interleaved"}
LAYOUT_COUNT = 4


class _DeviceBytes:
    """`nbytes` of device memory at `ptr` for torch.as_tensor (zero copy)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class ColumnPair:
    """An (input, aux, output) triple of device buffers from fl_column_pair_alloc: `layout` "separate" (one allocation each), "zoned"
    (one allocation, the output centred on a 64-GiB multiple: pins ~64 GiB), "interleaved" (round 6: built from 1-GiB physical chunks
    whose class of memory was measured -- the input inside one class, the output's chunks arranged so that the eight XCDs' write positions spread over classes) or "auto" (every
    candidate tried, a bare stream of in_bytes : out_bytes timed on each, the fastest kept -- synchronous, contents unspecified).
    .input / .aux / .output are uint8 torch tensors over the memory (zero copy; they do NOT own it), .layout the layout kept,
    .probe_GBps {"interleaved": .., "separate": .., "zoned": ..} (auto only), .classes the measured class of every chunk of an
    interleaved pair ("AAAAAAAAABBCCBBCC..", input first).  The memory is released by .free() or when this object dies: keep it for as
    long as the tensors are in use."""

    def __init__(self, in_bytes, out_bytes, device, aux_bytes=0, layout="auto", stream=None):
        import ctypes
        import torch
        from . import _lib
        self._lib = _lib.load()
        P = ctypes.c_void_p
        i, a, o, h = P(), P(), P(), P()
        kept = ctypes.c_int(-1)
        gbps = (ctypes.c_uint32 * LAYOUT_COUNT)()
        dev = torch.device(device)
        with torch.cuda.device(dev):
            st = P(torch.cuda.current_stream(dev).cuda_stream) if stream is None else stream
            rc = self._lib.fl_column_pair_alloc(in_bytes, aux_bytes, out_bytes, LAYOUTS[layout], st, ctypes.byref(i), ctypes.byref(a),
                                                ctypes.byref(o), ctypes.byref(h), ctypes.byref(kept), gbps)
        if rc != 0:
            from .codec import FastLanesError
            raise FastLanesError(rc, "fl_column_pair_alloc")
        self._handle = h
        self.layout = LAYOUT_NAMES[kept.value]
        self.probe_GBps = {LAYOUT_NAMES[k]: int(gbps[k]) for k in LAYOUT_NAMES if gbps[k]} if layout == "auto" else None
        self.classes = (self._lib.fl_internal_column_pair_classes(h) or b"").decode()

        def view(ptr, nbytes):
            if not nbytes:
                return torch.empty(0, dtype=torch.uint8, device=dev)
            return torch.as_tensor(_DeviceBytes(ptr.value, nbytes), device=dev)
        self.input, self.aux, self.output = view(i, in_bytes), view(a, aux_bytes), view(o, out_bytes)

    def free(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self.input = self.aux = self.output = None
            self._lib.fl_column_pair_free(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
