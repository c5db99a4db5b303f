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
        # round 6: the buffers of each DIRECTION in a constructed layout (fl_column_pair_alloc: FL_LAYOUT_INTERLEAVED), one pair per
        # direction and element type, sized for the type's largest case and sliced per width -- in memory of one class every variant
        # reads the same figure (the memory is the bound) and the sweep decides nothing
        from fastlanes_amd import placement as pl
        if CONSTRUCTED.get("ty") != ty:
            for q in CONSTRUCTED.get("pairs", []):
                q.free()
            torch.cuda.empty_cache()
            n_of = lambda w: min(10_000_000, (GB << 30) // (128 * w + 128 * T + 128))
            n_max = n_of(0)
            pk_max = max(n_of(w) * 128 * w for w in range(T + 1))
            dec = pl.ColumnPair(pk_max, n_max * 128 * T, dev, aux_bytes=n_max * 128, layout="interleaved")
            enc = pl.ColumnPair(n_max * 128 * T, 2 * pk_max, dev, aux_bytes=n_max * 128, layout="interleaved")
            print(f"# {ty}: constructed pairs, measured classes (input + bases first): decode {dec.classes} | encode {enc.classes}", flush=True)
            dec.input.copy_(rand_u8(pk_max, 2, dev))
            dec.aux.copy_(rand_u8(n_max * 128, 3, dev))
            enc.input.copy_(rand_u8(n_max * 128 * T, 1, dev))
            enc.aux.copy_(dec.aux)
            CONSTRUCTED.update(ty=ty, pairs=[dec, enc], pk_max=pk_max)
        dec, enc = CONSTRUCTED["pairs"]
        pk, out, bases = dec.input[:n * 128 * W].view(tdt), dec.output[:n * 128 * T].view(tdt), dec.aux[:n * 128].view(tdt)
        un, bases_enc = enc.input[:n * 128 * T].view(tdt), enc.aux[:n * 128].view(tdt)
        pk_out_c, pk_for_c = enc.output[:n * 128 * W].view(tdt), enc.output[CONSTRUCTED["pk_max"]:][:n * 128 * W].view(tdt)
    else:
        pk = rand_u8(n * 128 * W, 2, dev).view(tdt)
        un = rand_u8(n * 1024 * esz, 1, dev).view(tdt)
        out = torch.empty(n * 1024, dtype=tdt, device=dev)
        bases = rand_u8(n * 128, 3, dev).view(tdt)
        bases_enc = bases
        pk_out_c = pk_for_c = None
    refs = bases[:n]
    ops = {}
    if "--for" in sys.argv or ALL:      # FoR's bodies (rows of their own in the dispatch table since round 5)
        pk_for = torch.empty_like(pk) if pk_for_c is None else pk_for_c
        ops["unfor_pack"] = (lambda: fl.FoR.unfor_pack(W, pk, refs, output=out), n * (128 * W + 128 * T))
        ops["for_pack"] = (lambda: fl.FoR.for_pack(W, un, refs, output=pk_for), n * (128 * W + 128 * T))
    ops.update({"undelta_pack": (lambda: fl.Delta.undelta_pack(W, pk, bases, output=out), n * (128 * W + 128 + 128 * T))})
    if True:
        pk_out = torch.empty_like(pk) if pk_out_c is None else pk_out_c
        ops["undelta_pack_untr"] = (lambda: fl.Delta.undelta_pack_untranspose(W, pk, bases, output=out), n * (128 * W + 128 + 128 * T))
        ops["transp_delta_pack"] = (lambda: fl.Delta.transpose_delta_pack(W, un, bases_enc, output=pk_out), n * (128 * W + 128 + 128 * T))
    if ty not in seen_plain:
        ops["transpose"] = (lambda: fl.Transpose.transpose(un, output=out), n * 256 * T)
        ops["untranspose"] = (lambda: fl.Transpose.untranspose(un, output=out), n * 256 * T)
    if ty not in seen_plain:
        seen_plain.add(ty)
        ops["undelta"] = (lambda: fl.Delta.undelta(un, bases, output=out), n * (256 * T + 128))
        ops["delta"] = (lambda: fl.Delta.delta(un, bases, output=out), n * (256 * T + 128))
    if T >= 32:                            # the two-blocks-per-wavefront form of the wide types' undelta_pack (a table entry 10 + waves)
        ops["undelta_pack_2b"] = ops["undelta_pack"]
    if "--only" in sys.argv:               # e.g. --only undelta_pack,undelta_pack_2b
        keep_ops = sys.argv[sys.argv.index("--only") + 1].split(",")
        ops = {k: v for k, v in ops.items() if k in keep_ops}
    for name, (f, nbytes) in ops.items():
        res_t = pk_out if name == "transp_delta_pack" else pk_for if name == "for_pack" else out
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
    pk_out = pk_for = res_t = pk_out_c = pk_for_c = None
    torch.cuda.empty_cache()
