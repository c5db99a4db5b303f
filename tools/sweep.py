#!/usr/bin/env python3
"""Throughput sweep over the kernel families (HIP-event timing, random data, HBM-resident).
    python tools/sweep.py [--gb 24] [--reps 7] [--cases all|quick]
Prints one line per (op, type, width): ms, GB/s (algorithmic bytes), fraction of
the 8 TB/s HBM peak, G ints/s."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlanes_amd as fl  # noqa: E402

ALLWIDTH_OPS = ("unpack", "pack", "unfor_pack", "for_pack", "undelta_pack", "undelta_pack_untranspose", "transpose_delta_pack")
WINDOW_AB = "--window-ab" in sys.argv
BARE = "--bare" in sys.argv or ("--cases" in sys.argv and sys.argv[sys.argv.index("--cases") + 1] == "allwidths")
PLACEMENT = sys.argv[sys.argv.index("--placement") + 1] if "--placement" in sys.argv else "zoned"
ESZ = {"u8": 1, "u16": 2, "u32": 4, "u64": 8}
TDT = {"u8": torch.uint8, "u16": torch.uint16, "u32": torch.uint32, "u64": torch.uint64}
dev = torch.device("cuda:0")


def rnd(nbytes, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    return torch.randint(-2**63, 2**63 - 1, ((nbytes + 7) // 8,), dtype=torch.int64, device=dev, generator=g).view(torch.uint8)[:nbytes]


def bytes_per_block(op, ty, w):
    T = ESZ[ty] * 8
    if op in ("pack", "unpack", "for_pack", "unfor_pack"):
        return 128 * w + 128 * T
    if op in ("undelta_pack", "undelta_pack_untranspose", "transpose_delta_pack"):
        return 128 * w + 128 + 128 * T
    if op == "unpack_block_sums":
        return 128 * w + 8
    if op == "unpack_compare":
        return 128 * w + 128
    if op == "block_min_max":
        return 128 * T + 2 * ESZ[ty]
    if op in ("delta", "undelta"):
        return 2 * 128 * T + 128
    return 2 * 128 * T  # transpose / untranspose


SLAB = {"t": None, "classes": None, "map": ""}


def the_slab(nbytes):
    """ONE allocation for the whole sweep, made before anything else and never freed: freeing and re-allocating per case hands the
    cases alternately a fresh and a fragmented piece of the device memory (the class maps of profiles/r03_sweep_consume.txt
    alternated between AAAABBBBBBBBACCC and ABBCAABCAACCCBCC), which showed up as every second row of a sweep being 5-8 % low --
    undelta against delta, untranspose against transpose.  The memory classes of its 8-GiB granules are measured once."""
    from fastlanes_amd import placement as pl
    if SLAB["t"] is None or SLAB["t"].numel() < nbytes:
        SLAB["t"] = None
        torch.cuda.empty_cache()
        free, _ = torch.cuda.mem_get_info(dev)
        want = max(nbytes, 160 << 30)
        size = min(want, free - (24 << 30)) // pl.GRANULE_BYTES * pl.GRANULE_BYTES
        if size < nbytes:
            return None
        SLAB["t"] = torch.empty(size, dtype=torch.uint8, device=dev)
        SLAB["classes"] = pl.granule_classes(SLAB["t"])
        SLAB["map"] = "".join("." if c is None else "ABC"[c] for c in SLAB["classes"][0])
        print(f"# one {size >> 30}-GiB allocation for every row; memory class of its 8-GiB granules: {SLAB['map']}", flush=True)
    return SLAB["t"]


def run(op, ty, w, gb, reps):
    """One op on one column.  The column's input and output are carved from the sweep's ONE allocation (the_slab): for the
    materialising kernels the input at offset 0 and the output centred on the first 64-GiB multiple behind it (where a fresh
    allocation's first boundary between memory classes lies: fastlanes_amd/placement.py); for the fused consumers the input in a
    run of 8-GiB granules of one class and the thin output in a granule of another, by the measured class map.  Separately
    allocated tensors share a class or not at the driver's whim, which moved every row of rounds 1-2 by up to 8 %;
    --placement separate restores that."""
    from fastlanes_amd import placement as pl
    T = ESZ[ty] * 8
    esz = ESZ[ty]
    bpb = bytes_per_block(op, ty, w)
    n = max(32, int(gb * 1e9 / bpb))
    packed_in = op in ("unpack", "unfor_pack", "undelta_pack", "unpack_block_sums", "unpack_compare", "undelta_pack_untranspose")
    in_bytes = n * 128 * w if packed_in else n * 128 * T
    if op in ("pack", "for_pack", "transpose_delta_pack"):
        out_bytes = n * 128 * w
    elif op == "unpack_block_sums":
        out_bytes = n * 8
    elif op == "unpack_compare":
        out_bytes = n * 128
    elif op == "block_min_max":
        out_bytes = 2 * n * esz
    else:
        out_bytes = n * 128 * T
    aux_bytes = n * 128 + n * esz          # Delta bases, then FoR references
    lib = fl.load()
    placed = ""
    consumer = op in ("unpack_block_sums", "unpack_compare", "block_min_max")
    total = None
    if PLACEMENT == "zoned":
        try:
            i_off, a_off, o_off, total = pl._layout(in_bytes, out_bytes, aux_bytes)
        except ValueError:
            total = None
    slab = the_slab(max(total, (in_bytes // pl.GRANULE_BYTES + 3) * pl.GRANULE_BYTES)) if total is not None else None
    pair = None
    if PLACEMENT == "interleaved" and not consumer:
        # a CONSTRUCTED pair per row (fl_column_pair_alloc(FL_LAYOUT_INTERLEAVED): the input + aux inside one class of memory, the output
        # rotating through the others by position); the library keeps the 1-GiB chunks between rows (fl_internal_pair_chunk_cache)
        lib.fl_internal_pair_chunk_cache(96)
        pair = pl.ColumnPair(in_bytes, out_bytes, dev, aux_bytes=aux_bytes, layout="interleaved")
        src8, aux8, dst8 = pair.input, pair.aux, pair.output
        placed = f"constructed pair {pair.classes}"
    elif slab is not None and consumer and out_bytes <= pl.GRANULE_BYTES - (1 << 30):
        # a thin write stream: input in a run of granules of one memory class, output in a granule of another (classes measured once
        # on the sweep's slab: fastlanes_amd/placement.py)
        cls, rates = SLAB["classes"]
        k = max(1, (in_bytes + pl.GRANULE_BYTES - 1) // pl.GRANULE_BYTES)
        start, best, one_class = pl.choose_granules(cls, rates, k)
        src8 = slab[start * pl.GRANULE_BYTES:][:in_bytes]
        dst8 = slab[best * pl.GRANULE_BYTES:][:out_bytes]
        aux8 = slab[:0]
        placed = f"input from granule {start}{'' if one_class else ' (SPANS classes)'}, output in {best} of {SLAB['map']}"
    elif slab is not None:
        src8, aux8, dst8 = slab[i_off:i_off + in_bytes], slab[a_off:a_off + aux_bytes], slab[o_off:o_off + out_bytes]
    else:
        src8 = torch.empty(in_bytes, dtype=torch.uint8, device=dev)
        aux8 = torch.empty(aux_bytes, dtype=torch.uint8, device=dev)
        dst8 = torch.empty(out_bytes, dtype=torch.uint8, device=dev)
    for t, seed in ((src8, 1), (aux8, 3)):
        if t.numel() & ~7:
            assert lib.fl_fill_random(t.data_ptr(), t.numel() & ~7, seed, None) == 0
    src = src8.view(TDT[ty])
    dst = dst8[:(out_bytes // esz) * esz].view(TDT[ty]) if op not in ("unpack_block_sums", "unpack_compare", "block_min_max") else None
    bases = aux8[:n * 128].view(TDT[ty]) if aux8.numel() else None
    refs = aux8[n * 128:n * 128 + n * esz].view(TDT[ty]) if aux8.numel() else None
    if op == "pack":
        f = lambda: fl.BitPacking.pack(w, src, output=dst)
    elif op == "unpack":
        f = lambda: fl.BitPacking.unpack(w, src, output=dst)
    elif op == "for_pack":
        f = lambda: fl.FoR.for_pack(w, src, refs, output=dst)
    elif op == "unfor_pack":
        f = lambda: fl.FoR.unfor_pack(w, src, refs, output=dst)
    elif op == "undelta_pack":
        f = lambda: fl.Delta.undelta_pack(w, src, bases, output=dst)
    elif op == "unpack_block_sums":
        sums = dst8[:n * 8].view(torch.int64)
        f = lambda: fl.BitPacking.unpack_block_sums(w, src, output=sums)
    elif op == "unpack_compare":
        # the mask is a thin WRITE stream inside a read stream: sharing a region of the device memory with the packed input costs
        # it 10-15 % (profiles/exp_thin_stream_r03.txt); placed like every other output (fastlanes_amd/placement.py)
        mask = dst8[:n * 128].view(torch.int32)
        f = lambda: fl.BitPacking.unpack_compare(w, src, "<", (1 << w) // 2, output=mask)
    elif op == "block_min_max":
        mm = (dst8[:n * esz].view(TDT[ty]), dst8[n * esz:2 * n * esz].view(TDT[ty]))
        f = lambda: fl.BitPacking.block_min_max(src, output=mm)
    elif op == "undelta_pack_untranspose":
        f = lambda: fl.Delta.undelta_pack_untranspose(w, src, bases, output=dst)
    elif op == "transpose_delta_pack":
        f = lambda: fl.Delta.transpose_delta_pack(w, src, bases, output=dst)
    elif op in ("delta", "undelta"):
        g = getattr(fl.Delta, op)
        f = lambda: g(src, bases, output=dst)
    else:
        g = getattr(fl.Transpose, op)
        f = lambda: g(src, output=dst)
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); b.synchronize()
        ms.append(a.elapsed_time(b))
    ms.sort()
    med = ms[len(ms) // 2]
    gbps = n * bpb / med / 1e6
    if WINDOW_AB:
        # the same buffers under the whole-column tile map of rounds 1-3 (window 31) and under 2^16-block windows, whatever the
        # library's own choice for this kernel is (fl_kernels.hpp: xcd_tile), round-robin
        big = {"u64": 20, "u32": 21, "u16": 22, "u8": 23}[ty]            # 8 GiB of unpacked blocks per window
        alt = {31: [], 16: [], big: []}
        for _ in range(reps):
            for wnd in alt:
                lib.fl_internal_set_kernel_policy(wnd << 25)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); f(); b.record(); b.synchronize()
                alt[wnd].append(a.elapsed_time(b))
        lib.fl_internal_set_kernel_policy(0)
        placed = (placed + " " if placed else "") + "whole-column map %.3f, 2^16-block windows %.3f, 8-GiB windows %.3f" % tuple(
            n * bpb / sorted(alt[k])[len(alt[k]) // 2] / 8e9 for k in (31, 16, big))
    bare = None
    if BARE and op in ALLWIDTH_OPS:
        # a bare stream of the same bytes per wavefront, same cache policy / occupancy / tile map as the kernel the dispatch runs, on the
        # SAME buffers (fl_internal_bare_stream; it overwrites the output, which nothing reads afterwards)
        import ctypes
        Z, I = ctypes.c_size_t, ctypes.c_int
        iu, au, ou, nt, wv, wn, bpu = Z(), Z(), Z(), I(), I(), I(), ctypes.c_uint()
        code = 1 if op in ("pack", "for_pack", "transpose_delta_pack") else 2 if op.startswith("undelta_pack") else 0
        if lib.fl_internal_bare_stream_shape(code, T, w, *[ctypes.byref(x) for x in (iu, au, ou, nt, wv, wn, bpu)]) == 0:
            if op == "transpose_delta_pack":              # pack's shape plus the bases on the read side
                au = Z(128 * bpu.value)
            nu = n // bpu.value
            if pair is not None and pair.classes:
                wn = I(31)                                 # inside a constructed pair the library launches under the whole-column tile map
            g = lambda: lib.fl_internal_bare_stream(src8.data_ptr(), iu.value, aux8.data_ptr() if au.value else None, au.value, dst8.data_ptr(), ou.value, nu,
                                                    nt.value, wv.value, wn.value, None)
            g(); g()
            torch.cuda.synchronize()
            bms = []
            for _ in range(reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); g(); b.record(); b.synchronize()
                bms.append(a.elapsed_time(b))
            bare = nu * (iu.value + au.value + ou.value) / sorted(bms)[len(bms) // 2] / 1e6
    if pair is not None:
        src = dst = bases = refs = src8 = aux8 = dst8 = None
        del f
        pair.free()
    return {"op": op, "ty": ty, "w": w, "n_blocks": n, "ms": round(med, 4), "GBps": round(gbps, 1),
            "frac": round(gbps / 8000, 4), "Gints": round(n * 1024 / med / 1e6, 1), "placed": placed,
            "bare_GBps": round(bare, 1) if bare else None, "of_bare": round(gbps / bare, 4) if bare else None}


def host_tier(reps=3):
    """PCIe-inclusive rate of the host-pointer tier (numpy in -> numpy out, staged through HBM)."""
    import time
    import numpy as np
    n = 65536
    pk = np.random.default_rng(1).integers(0, 2**32, size=n * 224, dtype=np.uint32)
    fl.BitPacking.unpack(7, pk)
    best = min((lambda t0: (fl.BitPacking.unpack(7, pk), time.perf_counter() - t0)[1])(time.perf_counter()) for _ in range(reps))
    print(f"host tier unpack u32 W=7 n={n}: {best * 1e3:.2f} ms  {n * 1024 / best / 1e9:.2f} Gint/s "
          f"({n * 4992 / best / 1e9:.2f} GB/s over PCIe incl. hipMalloc/hipFree and pageable copies)", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=24.0)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--cases", default="quick")
    ap.add_argument("--wmin", type=int, default=0, help="keep only widths >= this")
    ap.add_argument("--wmax", type=int, default=64, help="keep only widths <= this")
    ap.add_argument("--types", default="", help="keep only these element types of the chosen cases (allwidths with --placement interleaved: one "
                    "process per type -- a constructed pair's address ranges are never re-used within a process, 868 rows exhaust them)")
    ap.add_argument("--json", default=None)
    ap.add_argument("--placement", default="zoned", choices=("zoned", "separate", "interleaved"),
                    help="interleaved (every materialising row; pairs of 8 GiB and more, so --gb 12): the column's buffers from fl_column_pair_alloc(FL_LAYOUT_INTERLEAVED) -- packed sides in one "
                         "class of memory, the other side arranged for the eight XCDs' write positions")
    ap.add_argument("--window-ab", action="store_true", help="every row also under the whole-column tile map and under 2^16-block windows")
    ap.add_argument("--bare", action="store_true", help="pack / unpack / FoR / undelta_pack rows: also a bare stream of the row's bytes on the row's buffers (always on for allwidths)")
    ap.add_argument("--batch-all", action="store_true", help="--cases batch: every element type and the pack direction too")
    ap.add_argument("--batch-policies", default="", help="--cases batch: comma-separated kernel policies to time next to the default")
    args = ap.parse_args()
    cases = []
    if args.cases == "quick":
        cases = [("unpack", "u32", 7), ("pack", "u32", 7), ("unfor_pack", "u32", 7), ("for_pack", "u32", 7),
                 ("undelta_pack", "u32", 12), ("unpack", "u64", 17), ("pack", "u64", 17),
                 ("unpack", "u16", 3), ("pack", "u16", 3), ("unpack", "u8", 3), ("pack", "u8", 3),
                 ("undelta_pack", "u16", 9), ("undelta_pack", "u64", 20), ("undelta_pack", "u8", 4)]
        for ty in ("u8", "u16", "u32", "u64"):
            cases += [(op, ty, 0) for op in ("delta", "undelta", "transpose", "untranspose")]
    elif args.cases == "orig":
        cases = [(op, ty, 0) for ty in ("u8", "u16", "u32", "u64") for op in ("transpose", "untranspose")]
        cases += [("undelta_pack_untranspose", "u32", 12), ("undelta_pack_untranspose", "u64", 20),
                  ("undelta_pack_untranspose", "u16", 9), ("undelta_pack_untranspose", "u8", 4),
                  ("transpose_delta_pack", "u32", 12), ("transpose_delta_pack", "u64", 20),
                  ("transpose_delta_pack", "u16", 9), ("transpose_delta_pack", "u8", 4)]
    elif args.cases == "consume":
        cases = [("unpack_compare", "u32", 7), ("unpack_compare", "u32", 20), ("unpack_compare", "u64", 17),
                 ("unpack_compare", "u16", 3), ("unpack_compare", "u8", 3),
                 ("unpack_block_sums", "u32", 7), ("unpack_block_sums", "u32", 20), ("unpack_block_sums", "u64", 17),
                 ("unpack_block_sums", "u16", 3), ("unpack_block_sums", "u8", 3),
                 ("block_min_max", "u32", 0), ("block_min_max", "u64", 0), ("block_min_max", "u16", 0), ("block_min_max", "u8", 0)]
    elif args.cases == "fused":
        cases = [("undelta_pack", "u32", 12), ("undelta_pack_untranspose", "u32", 12), ("transpose_delta_pack", "u32", 12),
                 ("undelta_pack_untranspose", "u64", 20), ("transpose_delta_pack", "u64", 20),
                 ("undelta_pack_untranspose", "u16", 9), ("transpose_delta_pack", "u16", 9),
                 ("undelta_pack_untranspose", "u8", 4), ("transpose_delta_pack", "u8", 4)]
    elif args.cases == "allwidths":
        # EVERY (T, W) x {unpack, pack, FoR's two bodies, undelta_pack, the two fused transpose extensions} through the automatic dispatch
        # (the `match width` of bitpacking.rs:82-95 that every W must serve), one slab, class map printed; summarised per (op, T) at the end
        for ty in ("u8", "u16", "u32", "u64"):
            for w in range(1, ESZ[ty] * 8 + 1):
                cases += [(op, ty, w) for op in ALLWIDTH_OPS]
    elif args.cases == "widths":
        for ty in ("u8", "u16", "u32", "u64"):
            T = ESZ[ty] * 8
            for w in sorted({1, 2, 3, T // 4, T // 2, T - 1, T}):
                cases += [("unpack", ty, w), ("pack", ty, w)]
    if args.cases == "host":
        host_tier()
        return
    if args.cases == "small":
        # launch-bound regime: one call per small array (e.g. a 64 Ki-value chunk = 64 blocks)
        for nb in (1, 8, 64, 512, 4096, 32768, 262144):
            pk = rnd(nb * 896, 1).view(torch.uint32)
            out = torch.empty(nb * 1024, dtype=torch.uint32, device=dev)
            for _ in range(20):
                fl.BitPacking.unpack(7, pk, output=out)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 200
            a.record()
            for _ in range(reps):
                fl.BitPacking.unpack(7, pk, output=out)
            b.record(); b.synchronize()
            us = a.elapsed_time(b) * 1e3 / reps
            print(f"unpack u32 W=7 n_blocks={nb:>7d}: {us:9.2f} us per call (back-to-back on one stream)  "
                  f"{nb * 1024 / us / 1e3:9.2f} Gint/s  {nb * 4992 / us / 1e3:8.1f} GB/s", flush=True)
        # the same loop as a HIP GRAPH (1000 chunks of 64 blocks, one captured launch each, replayed) next to the batch entry that
        # decodes the same 1000 chunks in ONE launch: what capturing a launch-bound caller loop buys, and what the batch entry buys
        n_arr, nb = 1000, 64
        pk_all = rnd(n_arr * nb * 896, 1).view(torch.uint32)
        un_all = torch.empty(n_arr * nb * 1024, dtype=torch.uint32, device=dev)
        chunks = [pk_all[a * nb * 224:(a + 1) * nb * 224] for a in range(n_arr)]
        outs = [un_all[a * nb * 1024:(a + 1) * nb * 1024] for a in range(n_arr)]
        lib = fl.load()
        side = torch.cuda.Stream()

        def loop(stream_handle):
            for c, o in zip(chunks, outs):
                assert lib.fl_u32_unpack(7, c.data_ptr(), o.data_ptr(), nb, stream_handle) == 0

        def timed(f, reps=20):
            f(); torch.cuda.synchronize()
            ms = []
            for _ in range(reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); f(); b.record(); b.synchronize()
                ms.append(a.elapsed_time(b))
            return sorted(ms)[len(ms) // 2]

        import ctypes
        with torch.cuda.stream(side):
            direct = timed(lambda: loop(ctypes.c_void_p(side.cuda_stream)))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                loop(ctypes.c_void_p(side.cuda_stream))
            graph = timed(g.replay)
            batch = fl.Batch(chunks, outs, [7] * n_arr)
            one = timed(lambda: batch.unpack())
        for name, ms in (("one C-ABI call per chunk", direct), ("the same 1000 calls captured in a HIP graph, replayed", graph), ("fl_u32_unpack_batch: ONE launch", one)):
            print(f"1000 chunks x 64 blocks, unpack u32 W=7, {name:56s}: {ms * 1e3:9.1f} us  {ms * 1e3 / n_arr:7.3f} us per chunk  "
                  f"{n_arr * nb * 1024 / ms / 1e6:8.1f} Gint/s", flush=True)
        return
    if args.cases == "batch":
        # many small arrays per launch (fl_<ty>_unpack_batch / _pack_batch) next to one device-tier call per array: 10 000 chunks of
        # 64 blocks (64 Ki values, the chunk size of the callers SURVEY.md 8(b) names)
        lib = fl.load()
        pols = [int(x, 0) for x in args.batch_policies.split(",")] if args.batch_policies else []

        def interleaved(variants, reps):
            """median ms of every variant, timed ROUND-ROBIN (one launch of each per round): a variant timed alone right after an
            idle stretch runs at lower clocks than the ones after it -- round 3's batch-vs-contiguous gap was partly that"""
            for _ in range(3):
                for f in variants.values():
                    f()
            torch.cuda.synchronize()
            ms = {k: [] for k in variants}
            for _ in range(reps):
                for k, f in variants.items():
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(); f(); b.record(); b.synchronize()
                    ms[k].append(a.elapsed_time(b))
            return {k: sorted(v)[len(v) // 2] for k, v in ms.items()}

        cases = [("u32", 7, "unpack"), ("u32", 12, "unpack"), ("u32", 20, "unpack")]
        if args.batch_all:
            cases += [("u32", 7, "pack"), ("u32", 20, "pack"), ("u64", 17, "unpack"), ("u64", 17, "pack"), ("u16", 9, "unpack"),
                      ("u16", 9, "pack"), ("u8", 3, "unpack"), ("u8", 3, "pack")]
        for ty, w, op in cases:
            n_arr, nb = 10000, 64
            esz, T = ESZ[ty], ESZ[ty] * 8
            ppb, opb = nb * 1024 * w // T, nb * 1024
            pk_all = rnd(n_arr * ppb * esz, 1).view(TDT[ty])
            un_all = (rnd(n_arr * opb * esz, 2) if op == "pack" else torch.empty(n_arr * opb * esz, dtype=torch.uint8, device=dev)).view(TDT[ty])
            if op == "pack":
                pk_all = torch.empty_like(pk_all)
            packed = [pk_all[a * ppb:(a + 1) * ppb] for a in range(n_arr)]
            outs = [un_all[a * opb:(a + 1) * opb] for a in range(n_arr)]
            batch = fl.Batch(packed, outs, [w] * n_arr)
            run_batch = batch.unpack if op == "unpack" else batch.pack

            def with_policy(pol):
                def f():
                    lib.fl_internal_set_kernel_policy(pol)
                    run_batch()
                    lib.fl_internal_set_kernel_policy(0)
                return f
            # the yardstick: the same 640 000 blocks as ONE contiguous column through fl_<ty>_unpack / _pack, same buffers
            if op == "unpack":
                one = lambda: fl.BitPacking.unpack(w, pk_all, output=un_all)
            else:
                one = lambda: fl.BitPacking.pack(w, un_all, output=pk_all)
            variants = {"batch": run_batch, "contiguous": one}
            variants.update({pol: with_policy(pol) for pol in pols})
            med = interleaved(variants, max(args.reps, 15))
            t, tc = med["batch"], med["contiguous"]
            nbytes = n_arr * nb * (128 * w + 128 * T)
            for pol in pols:
                # A/B of the batch kernel's launch shape on the same buffers (fastlanes_amd_internal.h: policy = 2 + 256 * waves/SIMD
                # + 65536 * blocks per wavefront + 2^24 * prefetch)
                print(f"    policy waves={(pol >> 8) & 255} blocks/wave={(pol >> 16) & 255} prefetch={pol >> 24}: {med[pol]:8.4f} ms  "
                      f"{nbytes / med[pol] / 8e9:.3f} of peak", flush=True)
            one()
            want = (un_all if op == "unpack" else pk_all).clone()
            (un_all if op == "unpack" else pk_all).zero_()
            run_batch()
            same = torch.equal(want.view(torch.uint8), (un_all if op == "unpack" else pk_all).view(torch.uint8))
            # the same arrays as one call each (what a chunk-at-a-time caller does today), through the raw C ABI
            f = getattr(lib, f"fl_{ty}_{op}")
            ptrs = [((p.data_ptr(), o.data_ptr()) if op == "unpack" else (o.data_ptr(), p.data_ptr())) for p, o in zip(packed, outs)]
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            for src, dst in ptrs:
                f(w, src, dst, nb, None)
            b.record(); b.synchronize()
            t1 = a.elapsed_time(b)
            print(f"{op}_batch {ty} W={w}: {n_arr} arrays x {nb} blocks in one launch {t:8.4f} ms  {n_arr * nb * 1024 / t / 1e6:7.1f} Gint/s  "
                  f"{nbytes / t / 1e6:7.1f} GB/s ({nbytes / t / 8e9:.3f} of peak)  {'== one big ' + op if same else 'MISMATCH'} | "
                  f"the same blocks as one contiguous column, one call: {tc:8.4f} ms ({(t / tc - 1) * 100:+.1f} %) | one call per array: {t1:8.3f} ms  "
                  f"{n_arr * nb * 1024 / t1 / 1e6:7.1f} Gint/s  (x{t1 / t:.1f})", flush=True)
            del batch, pk_all, un_all, want, packed, outs
        if args.batch_all:
            # Delta's fused decode over the same shape (fl_<ty>_undelta_pack_batch), against the same blocks as one contiguous call
            for ty, w in (("u32", 12), ("u16", 9), ("u64", 20)):
                n_arr, nb = 10000, 64
                esz, T = ESZ[ty], ESZ[ty] * 8
                L = 1024 // T
                ppb, opb = nb * 1024 * w // T, nb * 1024
                pk_all = rnd(n_arr * ppb * esz, 1).view(TDT[ty])
                bs_all = rnd(n_arr * nb * 128, 3).view(TDT[ty])
                un_all = torch.empty(n_arr * opb, dtype=TDT[ty], device=dev)
                packed = [pk_all[a * ppb:(a + 1) * ppb] for a in range(n_arr)]
                bases = [bs_all[a * nb * L:(a + 1) * nb * L] for a in range(n_arr)]
                outs = [un_all[a * opb:(a + 1) * opb] for a in range(n_arr)]
                batch = fl.Batch(packed, outs, [w] * n_arr, bases=bases)
                one = lambda: fl.Delta.undelta_pack(w, pk_all, bs_all, output=un_all)
                med = interleaved({"batch": lambda: batch.undelta_pack(), "contiguous": one}, max(args.reps, 15))
                one()
                want = un_all.clone()
                un_all.zero_()
                batch.undelta_pack()
                same = torch.equal(want.view(torch.uint8), un_all.view(torch.uint8))
                t, tc = med["batch"], med["contiguous"]
                nbytes = n_arr * nb * (128 * w + 128 + 128 * T)
                print(f"undelta_pack_batch {ty} W={w}: {n_arr} arrays x {nb} blocks in one launch {t:8.4f} ms  {n_arr * nb * 1024 / t / 1e6:7.1f} Gint/s  "
                      f"{nbytes / t / 1e6:7.1f} GB/s ({nbytes / t / 8e9:.3f} of peak)  {'== one big undelta_pack' if same else 'MISMATCH'} | "
                      f"the same blocks as one contiguous column, one call: {tc:8.4f} ms ({(t / tc - 1) * 100:+.1f} %)", flush=True)
                del batch, pk_all, bs_all, un_all, want, packed, bases, outs
        return
    if args.cases == "refbench":
        # What the reference's own criterion benches time (besides benches/bitpacking.rs, which bench.py's headline and
        # cpu_baseline cover):
        #   benches/delta.rs:10-44      fused undelta_pack::<W> vs unpack::<W> followed by undelta (u16 W=9), one block
        #   benches/transpose.rs:8-19   transpose of one u16 block
        # here batched over a column (HBM-resident, GB/s of algorithmic bytes) and as ONE-BLOCK device-tier calls (us per
        # back-to-back call: launch-bound, the shape of the reference's bench).
        def timed(f, reps):
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            ms = []
            for _ in range(reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); f(); b.record(); b.synchronize()
                ms.append(a.elapsed_time(b))
            return sorted(ms)[len(ms) // 2]

        def per_call_us(f, reps=300):
            for _ in range(20):
                f()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                f()
            b.record(); b.synchronize()
            return a.elapsed_time(b) * 1e3 / reps

        for ty, w in (("u16", 9), ("u32", 12)):
            T = ESZ[ty] * 8
            n = int(args.gb * 1e9 / (128 * w + 128 + 2 * 128 * T))
            pk = rnd(n * 128 * w, 1).view(TDT[ty])
            bases = rnd(n * 128, 3).view(TDT[ty])
            out = torch.empty(n * 1024, dtype=TDT[ty], device=dev)
            tmp = torch.empty(n * 1024, dtype=TDT[ty], device=dev)
            fused = timed(lambda: fl.Delta.undelta_pack(w, pk, bases, output=out), args.reps)
            want = out.clone()

            def unfused():
                fl.BitPacking.unpack(w, pk, output=tmp)
                fl.Delta.undelta(tmp, bases, output=out)
            unf = timed(unfused, args.reps)
            same = torch.equal(want.view(torch.uint8), out.view(torch.uint8))
            alg = n * (128 * w + 128 + 128 * T)
            print(f"benches/delta.rs shape, {ty} W={w}, {n} blocks: fused undelta_pack {fused:8.4f} ms ({alg / fused / 1e6:7.1f} GB/s, "
                  f"{n * 1024 / fused / 1e6:7.1f} Gint/s)  unpack + undelta {unf:8.4f} ms ({n * 1024 / unf / 1e6:7.1f} Gint/s)  "
                  f"speed-up {unf / fused:.2f}x  results {'identical' if same else 'DIFFER'}", flush=True)
            pk1, b1, o1, t1 = pk[:128 * w // ESZ[ty]], bases[:128 // ESZ[ty]], out[:1024], tmp[:1024]
            f1 = per_call_us(lambda: fl.Delta.undelta_pack(w, pk1, b1, output=o1))

            def unfused1():
                fl.BitPacking.unpack(w, pk1, output=t1)
                fl.Delta.undelta(t1, b1, output=o1)
            u1 = per_call_us(unfused1)
            print(f"    one block, device tier: fused {f1:6.2f} us per call, unpack + undelta {u1:6.2f} us  (speed-up {u1 / f1:.2f}x; launch-bound)", flush=True)
            del pk, bases, out, tmp, want
            torch.cuda.empty_cache()
        for ty in ("u16",):
            T = ESZ[ty] * 8
            n = int(args.gb * 1e9 / (2 * 128 * T))
            src = rnd(n * 128 * T, 1).view(TDT[ty])
            dst = torch.empty_like(src)
            for name, g in (("transpose", fl.Transpose.transpose), ("untranspose", fl.Transpose.untranspose)):
                ms = timed(lambda: g(src, output=dst), args.reps)
                s1, d1 = src[:1024], dst[:1024]
                us = per_call_us(lambda: g(s1, output=d1))
                print(f"benches/transpose.rs shape, {name} {ty}: {n} blocks {ms:8.4f} ms ({n * 2 * 128 * T / ms / 1e6:7.1f} GB/s, "
                      f"{n * 1024 / ms / 1e6:7.1f} Gint/s); one block, device tier: {us:6.2f} us per call", flush=True)
        return
    if args.cases == "mixed":
        # FoR's and Delta's bodies over device-resident mixed-width columns (fl_<ty>_unfor_pack_widths, ..) next to plain
        # unpack_widths / pack_widths of the same column, and an encoder's whole chain: block_min_max -> for_widths ->
        # widths_to_offsets -> for_pack_widths (two passes over the values).  Widths seeded-random in 1..T-1, separate tensors.
        lib = fl.load()
        for ty in ("u32", "u64", "u16", "u8"):
            T, esz = ESZ[ty] * 8, ESZ[ty]
            L = 1024 // T
            n = max(64, int(args.gb * 1e9 / (128 * T * 1.5)))
            g = torch.Generator(device=dev); g.manual_seed(31 + T)
            widths = torch.randint(1, T, (n,), dtype=torch.int64, device=dev, generator=g).to(torch.uint8)
            offsets, total = fl.widths_to_offsets(ty, widths)
            pbytes = int(total.item())
            pair = None
            if PLACEMENT == "interleaved":
                from fastlanes_amd import placement as pl
                # one constructed pair per DIRECTION: the decoders read `col` (one class) and write `un` (the other two), the encoders
                # read `un_enc` (one class) and write `back` (the other two)
                pair = pl.ColumnPair(pbytes, n * 128 * T, dev, aux_bytes=n * 128, layout="interleaved")
                pair_enc = pl.ColumnPair(n * 128 * T, pbytes, dev, aux_bytes=n * 128, layout="interleaved")
                print(f"# {ty}: constructed pairs, measured classes (input + bases first): decode {pair.classes} | encode {pair_enc.classes}", flush=True)
                col, un = pair.input.view(TDT[ty]), pair.output.view(TDT[ty])
                un_enc, back = pair_enc.input.view(TDT[ty]), pair_enc.output.view(TDT[ty])
                col.view(torch.uint8).copy_(rnd(pbytes, 1))
                bases = pair.aux.view(TDT[ty])
                bases.view(torch.uint8).copy_(rnd(n * 128, 3))
                bases_enc = pair_enc.aux.view(TDT[ty])
                bases_enc.copy_(bases)
            else:
                col = rnd(pbytes, 1).view(TDT[ty])
                bases = rnd(n * 128, 3).view(TDT[ty])
                un = torch.empty(n * 1024, dtype=TDT[ty], device=dev)
                back = torch.empty_like(col)
                un_enc, bases_enc, pair_enc = un, bases, None
            # references with the top bit clear: reference + field never wraps, so the encoder below finds widths <= the decoder's
            refs = (rnd(n * 8, 2).view(torch.int64) & ((1 << (T - 1)) - 1)).view(torch.uint8).view(-1, 8)[:, :esz].contiguous().view(TDT[ty]).reshape(-1)
            mm = (torch.empty(n, dtype=TDT[ty], device=dev), torch.empty(n, dtype=TDT[ty], device=dev))
            fl.unfor_pack_widths(widths, offsets, col, refs, output=un_enc)       # the values every encoder row below reads

            def encoder_chain():
                lo, hi = fl.BitPacking.block_min_max(un_enc, output=mm)
                w2 = fl.for_widths(lo, hi)
                o2, _ = fl.widths_to_offsets(ty, w2)
                fl.for_pack_widths(w2, o2, un_enc, lo, back, check=False)

            two_sided = pbytes + n * 128 * T
            rows = (
                ("unpack_widths", two_sided, lambda: fl.unpack_widths(widths, offsets, col, output=un, check=False)),
                ("unfor_pack_widths", two_sided, lambda: fl.unfor_pack_widths(widths, offsets, col, refs, output=un, check=False)),
                ("undelta_pack_widths", two_sided + n * 128, lambda: fl.undelta_pack_widths(widths, offsets, col, bases, output=un, check=False)),
                ("undelta_pack_untranspose_widths", two_sided + n * 128,
                 lambda: fl.undelta_pack_widths(widths, offsets, col, bases, output=un, check=False, untranspose=True)),
                ("restore", 0, lambda: fl.unfor_pack_widths(widths, offsets, col, refs, output=un_enc, check=False)),
                ("pack_widths", two_sided, lambda: fl.pack_widths(widths, offsets, un_enc, back, check=False)),
                ("for_pack_widths", two_sided, lambda: fl.for_pack_widths(widths, offsets, un_enc, refs, back, check=False)),
                ("transpose_delta_pack_widths", two_sided + n * 128, lambda: fl.transpose_delta_pack_widths(widths, offsets, un_enc, bases_enc, back, check=False)),
                ("FoR encoder chain (4 launches)", 2 * n * 128 * T + pbytes, encoder_chain),
            )
            for name, nbytes, f in rows:
                f(); f()
                torch.cuda.synchronize()
                if not nbytes:
                    continue
                ms = []
                for _ in range(args.reps):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(); f(); b.record(); b.synchronize()
                    ms.append(a.elapsed_time(b))
                med = sorted(ms)[len(ms) // 2]
                print(f"{name:32s} {ty:4s} n={n:>9d} {med:9.4f} ms {nbytes / med / 1e6:8.1f} GB/s {nbytes / med / 8e9:.3f} "
                      f"{n * 1024 / med / 1e6:8.1f} Gint/s", flush=True)
            del col, un, back, refs, bases, mm, un_enc, bases_enc
            if pair is not None:
                pair.free()
                pair_enc.free()
            torch.cuda.empty_cache()
        return
    if args.cases == "single":
        # batched unpack_single (bitpacking.rs:132-200; benches/bitpacking.rs:36-65 times one lookup): k lookups into an n-block column
        # -- random, sorted, strided (one per block: every lookup a different block) and dense (all 1024 of consecutive blocks);
        # uniform-width and mixed-width entry points.  A lookup needs 1-2 words of sizeof(T) bytes (:164-178); the memory system
        # moves 32-byte sectors (64-byte requests on gfx950), so a RANDOM lookup costs a sector however small T is.
        for ty, w in (("u32", 7), ("u64", 17), ("u16", 3), ("u8", 3)):
            esz, T = ESZ[ty], ESZ[ty] * 8
            n, k = 1_000_000, 64_000_000
            pk = rnd(n * 128 * w, 1).view(TDT[ty])
            g = torch.Generator(device=dev); g.manual_seed(5)
            idx = torch.randint(0, n * 1024, (k,), dtype=torch.int64, device=dev, generator=g)
            out1 = torch.empty(k, dtype=TDT[ty], device=dev)
            widths = torch.full((n,), w, dtype=torch.uint8, device=dev)
            offsets, _ = fl.widths_to_offsets(ty, widths)
            patterns = (("random", idx), ("sorted", torch.sort(idx).values),
                        ("strided (one per block)", (torch.arange(k, dtype=torch.int64, device=dev) % n) * 1024 + (torch.arange(k, dtype=torch.int64, device=dev) * 7) % 1024),
                        ("dense (whole blocks in order)", torch.arange(k, dtype=torch.int64, device=dev)))
            lib = fl.load()
            err = torch.zeros(1, dtype=torch.int32, device=dev)
            for name, ii in patterns:
                # the raw C ABI: no allocation and no error-flag read-back inside the timed region
                for label, f in (("unpack_single", lambda: getattr(lib, f"fl_{ty}_unpack_single")(w, pk.data_ptr(), n, ii.data_ptr(), k, out1.data_ptr(), err.data_ptr(), None)),
                                 ("unpack_single_widths", lambda: getattr(lib, f"fl_{ty}_unpack_single_widths")(widths.data_ptr(), offsets.data_ptr(), pk.data_ptr(), n * 128 * w, n, ii.data_ptr(), k, out1.data_ptr(), err.data_ptr(), None))):
                    assert f() == 0 and f() == 0
                    torch.cuda.synchronize()
                    ms = []
                    for _ in range(args.reps):
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record(); f(); b.record(); b.synchronize()
                        ms.append(a.elapsed_time(b))
                    t = sorted(ms)[len(ms) // 2]
                    io = k * (8 + esz)                  # the index read and the value written, per lookup
                    print(f"{label:21s} {ty:4s} W={w:<2d} {k} lookups, {name:30s} {t:8.3f} ms  {k / t / 1e6:7.2f} G lookups/s  "
                          f"index+result stream {io / t / 1e6:7.1f} GB/s ({io / t / 8e9:.3f} of peak)", flush=True)
            del pk, idx, out1, widths, offsets
            torch.cuda.empty_cache()
        return
    out = []
    if args.types:
        cases = [c for c in cases if c[1] in args.types.split(",")]
    cases = [c for c in cases if args.wmin <= c[2] <= args.wmax]
    for op, ty, w in cases:
        r = run(op, ty, w, args.gb, args.reps)
        out.append(r)
        print(f"{op:{24 if args.cases == 'allwidths' else 13}s} {ty:4s} W={w:<3d} n={r['n_blocks']:>9d} {r['ms']:9.4f} ms {r['GBps']:8.1f} GB/s {r['frac']:.3f} {r['Gints']:8.1f} Gint/s" +
              (f"   bare stream {r['bare_GBps']:7.1f} GB/s -> {r['of_bare']:.3f} of it" if r.get("of_bare") else "") + (f"   [{r['placed']}]" if r.get("placed") else ""), flush=True)
        torch.cuda.empty_cache()
    if args.cases == "allwidths":
        print("# ---- summary: fraction of the 8 TB/s peak per (op, type) over all widths 1..T: min (at W) / median / max (at W)")
        for op in ALLWIDTH_OPS:
            for ty in ("u8", "u16", "u32", "u64"):
                rows = sorted((r["frac"], r["w"]) for r in out if r["op"] == op and r["ty"] == ty)
                if not rows:
                    continue
                ob = sorted((r["of_bare"], r["w"]) for r in out if r["op"] == op and r["ty"] == ty and r.get("of_bare"))
                print(f"# {op:24s} {ty:4s} min {rows[0][0]:.3f} (W={rows[0][1]:<2d})  median {rows[len(rows) // 2][0]:.3f}  max {rows[-1][0]:.3f} (W={rows[-1][1]:<2d})" +
                      (f"   | of the bare stream of the same bytes on the same buffers: min {ob[0][0]:.3f} (W={ob[0][1]:<2d})  median {ob[len(ob) // 2][0]:.3f}" if ob else ""))
        worst = sorted(out, key=lambda r: r["frac"])[:8]
        print("# ---- the eight slowest (op, T, W): " + "; ".join(f"{r['op']} {r['ty']} W={r['w']} {r['frac']:.3f}" + (f" ({r['of_bare']:.2f} of its bare stream)" if r.get("of_bare") else "") for r in worst))
        wb = sorted((r for r in out if r.get("of_bare")), key=lambda r: r["of_bare"])[:8]
        print("# ---- the eight furthest below their own bare stream: " + "; ".join(f"{r['op']} {r['ty']} W={r['w']} {r['of_bare']:.3f} (frac {r['frac']:.3f})" for r in wb))
    if args.json:
        json.dump(out, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
