#!/usr/bin/env python3
"""A/B on the GPU box: Delta's kernels (undelta_pack, undelta, delta) through the cell-column kernels vs the
wave-per-block chain kernels (fl_chain.hpp) at several occupancies, interleaved on the SAME buffers via
fl_internal_set_kernel_policy (1 = cell-column; 2 + 256*n = wave-per-block at n waves/SIMD).  GB/s of algorithmic bytes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastlanes_amd as fl  # noqa: E402
from bench import rand_u8  # noqa: E402

lib = fl.load()
dev = torch.device("cuda", 0)
TD = {"u8": (torch.uint8, 8), "u16": (torch.uint16, 16), "u32": (torch.uint32, 32), "u64": (torch.uint64, 64)}
WAVES = (3, 4, 5, 6, 8)
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3
cases = [("u32", 12), ("u32", 7), ("u32", 3), ("u32", 20), ("u32", 28), ("u64", 20), ("u64", 5), ("u64", 40), ("u64", 60),
         ("u16", 9), ("u16", 3), ("u16", 14), ("u8", 4), ("u8", 7)]
ALL = "--all" in sys.argv          # every width of every type: the basis of tools/make_dispatch.py
GB = 12
if ALL:
    cases = [(ty, w) for ty in ("u32", "u64", "u16", "u8") for w in range(TD[ty][1] + 1)]
    GB = 6
if "--types" in sys.argv:          # e.g. --types u16,u8: re-sweep some element types only (round 4: the narrow types' chain kernels
    keep = sys.argv[sys.argv.index("--types") + 1].split(",")   # now keep several blocks in flight per wavefront)
    cases = [c for c in cases if c[0] in keep]
if "--start-at" in sys.argv:       # e.g. --start-at u64:64: resume a sweep that died (order: u32, u64, u16, u8, widths ascending)
    ty0, w0 = sys.argv[sys.argv.index("--start-at") + 1].split(":")
    cases = cases[cases.index((ty0, int(w0))):]
if "--gb" in sys.argv:             # bytes moved per launch; BASELINE's config 4 is a 57.6 GB launch
    GB = int(sys.argv[sys.argv.index("--gb") + 1])
print("GB/s, median of %d; cc = cell-column, then wave-per-block at %s waves/SIMD" % (ROUNDS, " ".join(map(str, WAVES))))
seen_plain = set()
CONSTRUCTED = {}
for ty, W in cases:
    tdt, T = TD[ty]
    esz = T // 8
    n = min(10_000_000, (GB << 30) // (128 * W + 128 * T + 128))
    if "--constructed" in sys.argv:
        # round 6: the buffers of each DIRECTION in a constructed pair (fl_column_pair_alloc: FL_LAYOUT_INTERLEAVED) -- in memory of one
        # class every variant reads the same figure (the memory is the bound) and the sweep decides nothing.  ONE PAIR PER ROW AND
        # DIRECTION, exactly the row's size (second half of round 6: a pair's output is arranged for the eight write positions of ITS
        # length -- a slice of a larger pair is not a constructed layout); the library keeps the 1-GiB chunks between rows.  Bases, then
        # references, in the pair's aux buffer.  One process per element type (--types): address ranges are never re-used.
        from fastlanes_amd import placement as pl
        lib.fl_internal_pair_chunk_cache(128)
        for q in CONSTRUCTED.get("pairs", []):
            q.free()
        aux_b = n * 128 + ((n * esz + 255) & ~255)
        dec = pl.ColumnPair(max(n * 128 * W, 256), n * 128 * T, dev, aux_bytes=aux_b, layout="interleaved")
        enc = pl.ColumnPair(n * 128 * T, max(n * 128 * W, 256), dev, aux_bytes=aux_b, layout="interleaved")
        pairs = [dec, enc]
        flat = None
        if ty not in seen_plain:                           # transposes / delta / undelta: unpacked -> unpacked
            flat = pl.ColumnPair(n * 128 * T, n * 128 * T, dev, aux_bytes=n * 128, layout="interleaved")
            pairs.append(flat)
        print(f"# {ty} W={W}: constructed pairs, measured classes (input + aux first): decode {dec.classes} | encode {enc.classes}", flush=True)
        for q in pairs:
            assert lib.fl_fill_random(q.input.data_ptr(), q.input.numel() & ~7, 2, None) == 0
            assert lib.fl_fill_random(q.aux.data_ptr(), q.aux.numel() & ~7, 3, None) == 0
        CONSTRUCTED.update(ty=ty, pairs=pairs)
        pk, out, bases = dec.input[:n * 128 * W].view(tdt), dec.output[:n * 128 * T].view(tdt), dec.aux[:n * 128].view(tdt)
        refs_dec = dec.aux[n * 128:n * 128 + n * esz].view(tdt)
        un, bases_enc = enc.input[:n * 128 * T].view(tdt), enc.aux[:n * 128].view(tdt)
        refs_enc = enc.aux[n * 128:n * 128 + n * esz].view(tdt)
        pk_out_c = pk_for_c = enc.output[:n * 128 * W].view(tdt)
    else:
        pk = rand_u8(n * 128 * W, 2, dev).view(tdt)
        un = rand_u8(n * 1024 * esz, 1, dev).view(tdt)
        out = torch.empty(n * 1024, dtype=tdt, device=dev)
        bases = rand_u8(n * 128, 3, dev).view(tdt)
        bases_enc = bases
        pk_out_c = pk_for_c = None
    refs = bases[:n]
    refs_e = refs
    flat_un = flat_out = flat_bases = None
    if "--constructed" in sys.argv:
        refs, refs_e = refs_dec, refs_enc
        if CONSTRUCTED["pairs"][-1] is not enc:
            fq = CONSTRUCTED["pairs"][-1]
            flat_un, flat_out, flat_bases = fq.input.view(tdt), fq.output.view(tdt), fq.aux[:n * 128].view(tdt)
    ops = {}
    if "--for" in sys.argv or ALL:      # FoR's bodies (rows of their own in the dispatch table since round 5)
        pk_for = torch.empty_like(pk) if pk_for_c is None else pk_for_c
        ops["unfor_pack"] = (lambda: fl.FoR.unfor_pack(W, pk, refs, output=out), n * (128 * W + 128 * T))
        ops["for_pack"] = (lambda: fl.FoR.for_pack(W, un, refs_e, output=pk_for), n * (128 * W + 128 * T))
    ops.update({"undelta_pack": (lambda: fl.Delta.undelta_pack(W, pk, bases, output=out), n * (128 * W + 128 + 128 * T))})
    if True:
        pk_out = torch.empty_like(pk) if pk_out_c is None else pk_out_c
        ops["undelta_pack_untr"] = (lambda: fl.Delta.undelta_pack_untranspose(W, pk, bases, output=out), n * (128 * W + 128 + 128 * T))
        ops["transp_delta_pack"] = (lambda: fl.Delta.transpose_delta_pack(W, un, bases_enc, output=pk_out), n * (128 * W + 128 + 128 * T))
    if ty not in seen_plain:
        seen_plain.add(ty)
        f_un, f_out, f_bases = (un, out, bases) if flat_un is None else (flat_un, flat_out, flat_bases)
        ops["transpose"] = (lambda: fl.Transpose.transpose(f_un, output=f_out), n * 256 * T)
        ops["untranspose"] = (lambda: fl.Transpose.untranspose(f_un, output=f_out), n * 256 * T)
        ops["undelta"] = (lambda: fl.Delta.undelta(f_un, f_bases, output=f_out), n * (256 * T + 128))
        ops["delta"] = (lambda: fl.Delta.delta(f_un, f_bases, output=f_out), n * (256 * T + 128))
    if T >= 32:                            # the two-blocks-per-wavefront form of the wide types' undelta_pack (a table entry 10 + waves)
        ops["undelta_pack_2b"] = ops["undelta_pack"]
    if "--only" in sys.argv:               # e.g. --only undelta_pack,undelta_pack_2b
        keep_ops = sys.argv[sys.argv.index("--only") + 1].split(",")
        ops = {k: v for k, v in ops.items() if k in keep_ops}
    for name, (f, nbytes) in ops.items():
        res_t = pk_out if name == "transp_delta_pack" else pk_for if name == "for_pack" else out
        if flat_out is not None and name in ("transpose", "untranspose", "undelta", "delta"):
            res_t = flat_out
        two = 65536 * 2 if name.endswith("_2b") else 0
        lib.fl_internal_set_kernel_policy(1)
        f()
        ref = res_t.clone()
        lib.fl_internal_set_kernel_policy(2 + two)
        f()
        same = torch.equal(ref.view(torch.uint8), res_t.view(torch.uint8))
        del ref
        res = {}
        pols = [1] + [2 + 256 * w + two for w in WAVES]
        for _ in range(ROUNDS):
            for p in pols:
                lib.fl_internal_set_kernel_policy(p)
                f()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); f(); b.record()
                torch.cuda.synchronize()
                res.setdefault(p, []).append(a.elapsed_time(b))
        g = [nbytes / sorted(res[p])[len(res[p]) // 2] / 1e6 for p in pols]
        print(f"{ty:3s} W={W:<2d} {name:17s}{'' if same else ' MISMATCH'} | cc {g[0]:6.0f}  wpb " + " ".join(f"{x:6.0f}" for x in g[1:]), flush=True)
    lib.fl_internal_set_kernel_policy(0)
    # the lambdas in `ops` hold every buffer of the case: drop them with the buffers, or the caching allocator fragments until a
    # 45-GB case no longer fits (round 5: both boxes died at u64 W=64)
    del ops, f, pk, un, out, bases, refs, bases_enc
    refs_e = refs_dec = refs_enc = flat_un = flat_out = flat_bases = f_un = f_out = f_bases = None
    pk_out = pk_for = res_t = pk_out_c = pk_for_c = None
    torch.cuda.empty_cache()
