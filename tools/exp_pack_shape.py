"""Round 6: the ENCODE direction of the wide types (pack / pack_widths u32, u64) in a constructed layout: waves per SIMD x blocks
per wavefront (prefetched by LDS-DMA) x tile-map window, same buffers, policies round-robin.  The narrow types' encode got +6 % from
two blocks in flight per wavefront (profiles/r06_exp_narrow_bpw.txt: u8 pack_widths 0.849 where u32's sits at 0.81 on the same
read : write proportion); this asks whether the wide types' one-block wavefront is a bound of the same kind.
    python tools/exp_pack_shape.py [cases: u32w7,u64w17,u32mixed,u64mixed] [--layout interleaved|separate] [--op pack|unpack|undelta_pack] [--bpw 1,2,4]
(--op unpack: the same matrix for the decode direction -- the narrow types' mixed-width kernels want every wave slot, the memory wants fewer)"""
import sys, os, statistics, torch
sys.path.insert(0, os.getcwd())
import fastlanes_amd as fl
from fastlanes_amd import placement as pl

lib = fl.load(); dev = torch.device("cuda:0")
TDT = {"u8": torch.uint8, "u16": torch.uint16, "u32": torch.uint32, "u64": torch.uint64}
BITS = {"u8": 8, "u16": 16, "u32": 32, "u64": 64}
args = [a for a in sys.argv[1:] if not a.startswith("--")]
layout = "interleaved"
if "--layout" in sys.argv: layout = sys.argv[sys.argv.index("--layout") + 1]
cases = args[0].split(",") if args else ["u32w7", "u64w17", "u32mixed"]
WAVES = (3, 4, 5, 6, 8)
OP = sys.argv[sys.argv.index("--op") + 1] if "--op" in sys.argv else "pack"
BPW = [int(x) for x in sys.argv[sys.argv.index("--bpw") + 1].split(",")] if "--bpw" in sys.argv else [1, 2, 4]
WINS = (0, 31, 14) if OP == "pack" else (0,)


def policies(T):
    pols = [("default", 0)]
    for win in WINS:
        for b, p in [(b, int(b > 1)) for b in BPW]:
            if 4 * b * 128 * T > 64 * 1024: continue
            for w in WAVES:
                if p and 4 * b * 128 * T * w > 160 * 1024: continue           # the images of `w` workgroups per CU would not fit
                pols.append((f"win{win:<2d} bpw{b}{'pf' if p else '  '} w{w}", 2 + 256 * w + 65536 * b + (1 << 24) * p + (1 << 25) * win))
    return pols


for case in cases:
    ty = case[:3] if case[2].isdigit() else case[:2]
    T = BITS[ty]
    mixed = case.endswith("mixed")
    if mixed:
        n = int(24e9 / (128 * T * 1.5))
        g = torch.Generator(device=dev); g.manual_seed(31 + T)
        widths = torch.randint(1, T, (n,), dtype=torch.int64, device=dev, generator=g).to(torch.uint8)
        offsets, total = fl.widths_to_offsets(ty, widths); pb = int(total)
    else:
        W = int(case.split("w")[1])
        n = int(40e9 / (128 * (T + W)))
        pb = n * 128 * W
    pair = pl.ColumnPair(n * 128 * T, pb, dev, layout=layout) if OP == "pack" else pl.ColumnPair(pb, n * 128 * T, dev, aux_bytes=n * 128, layout=layout)
    if OP == "undelta_pack":
        assert lib.fl_fill_random(pair.aux.data_ptr(), pair.aux.numel() & ~7, 6, None) == 0
        bases = pair.aux.view(TDT[ty])
    assert lib.fl_fill_random(pair.input.data_ptr(), pair.input.numel() & ~7, 5, None) == 0
    un, col = (pair.input.view(TDT[ty]), pair.output.view(TDT[ty])) if OP == "pack" else (pair.output.view(TDT[ty]), pair.input.view(TDT[ty]))
    pols = policies(T)
    res = {k: [] for k, _ in pols}
    for r in range(3):
        for name, pol in pols:
            lib.fl_internal_set_kernel_policy(pol)
            ms = []
            for i in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                if OP == "pack":
                    if mixed: fl.pack_widths(widths, offsets, un, col, check=False)
                    else: fl.BitPacking.pack(W, un, output=col)
                elif OP == "undelta_pack" and mixed: fl.undelta_pack_widths(widths, offsets, col, bases, output=un, check=False)
                elif OP == "undelta_pack": fl.Delta.undelta_pack(W, col, bases, output=un)
                elif mixed: fl.unpack_widths(widths, offsets, col, output=un, check=False)
                else: fl.BitPacking.unpack(W, col, output=un)
                b.record(); b.synchronize()
                if i: ms.append(a.elapsed_time(b))
            res[name].append((pb + n * 128 * T) / statistics.median(ms) / 8e9)
    lib.fl_internal_set_kernel_policy(0)
    print(f"{case} {OP} n={n} {layout} {pair.classes}")
    med = {k: statistics.median(v) for k, v in res.items()}
    print(f"   default            {med['default']:.3f}")
    for win in WINS:
        for b, p in [(b, int(b > 1)) for b in BPW]:
            row = [med.get(f"win{win:<2d} bpw{b}{'pf' if p else '  '} w{w}") for w in WAVES]
            if any(x is not None for x in row):
                print(f"   win{win:<2d} bpw{b}{'pf' if p else '  '}  " + "  ".join(f"w{w} {x:.3f}" if x is not None else f"w{w}   -  " for w, x in zip(WAVES, row)))
    pair.free()
    sys.stdout.flush()
