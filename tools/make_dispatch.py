#!/usr/bin/env python3
"""Regenerate fastlanes_amd/csrc/fl_dispatch_table.inc from A/B sweeps.

Two kernel designs compute identical bytes for most entry points: the per-(T,W) CELL-COLUMN kernels (fl_kernels.hpp)
and the runtime-width WAVE-PER-BLOCK kernels (fl_widths.hpp, fl_chain.hpp).  tools/abuniform (pack / unpack, every
(T, W)) and tools/abchain.py --all (Delta's kernels, the transposes and the fused transpose extensions) time both on the
SAME buffers, the wave-per-block one at 3 / 4 / 5 / 6 / 8 wavefronts per SIMD.  This script turns the sweeps of two or
more boxes into the table the library dispatches on:

    python tools/make_dispatch.py --uniform profiles/abuniform_r03a.txt profiles/abuniform_r03b.txt \\
                                  --chain profiles/abchain_r03a.txt profiles/abchain_r03b.txt [--margin 0.02] [--check]

Rule (stated in the generated file too).  For every (op, T, W):
  * k* = the occupancy whose wave-per-block rate relative to the cell-column kernel has the best geometric mean over
    the boxes;
  * the CELL-COLUMN kernel is chosen only if it leads wave-per-block@k* by more than `margin` on EVERY box;
  * otherwise the wave-per-block kernel at k* waves per SIMD (ties, and boxes that disagree, go to the one
    runtime-width kernel: it needs no per-(T,W) instance).
A table entry is the waves/SIMD to launch the wave-per-block kernel with, or 0 for the cell-column kernel; the per-(T,W)
cell-column instances whose entry is non-zero are not built at all (fl_kernels.hpp: cell_column_built).

--check regenerates in memory and fails if the committed table differs (tests/test_dispatch_table.py).
"""
import argparse
import math
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "fastlanes_amd", "csrc", "fl_dispatch_table.inc")
WAVES = (3, 4, 5, 6, 8)
TYPES = (8, 16, 32, 64)
# ops of the uniform sweep, ops of the chain sweep that depend on the width, and the per-type ones
UNIFORM_OPS = ("unpack", "pack")
CHAIN_W_OPS = {"undelta_pack": "undelta_pack", "undelta_pack_untr": "undelta_pack_untranspose",
               "transp_delta_pack": "transpose_delta_pack", "unfor_pack": "unfor_pack", "for_pack": "for_pack",
               "undelta_pack_2b": "undelta_pack_2b"}
# undelta_pack of u32 / u64 also exists with TWO blocks per wavefront in lockstep (fl_dispatch.hpp: TWO_BLOCKS): a table entry 10 + k
TWO_BLOCKS = 10
# FoR's rows (round 5): if NO chain file holds them (sweeps older than round 5) they follow unpack / pack, as the library did then
FOR_FOLLOWS = {"unfor_pack": "unpack", "for_pack": "pack"}
CHAIN_T_OPS = ("undelta", "delta", "untranspose", "transpose")

NUM = r"\s+(\d+)"
RE_UNIFORM = re.compile(r"^u(\d+)\s+W=(\d+)\s*([A-Z\- ]*)\| unpack cc" + NUM + r"\s+wpb" + NUM * 5 + r" \| pack cc" + NUM + r"\s+wpb" + NUM * 5)
RE_CHAIN = re.compile(r"^u(\d+)\s+W=(\d+)\s+(\w+)\s*(MISMATCH)?\s*\| cc" + NUM + r"\s+wpb" + NUM * 5)


def parse_uniform(path):
    """{(op, T, W): (cc, [wpb at WAVES])}"""
    out = {}
    for line in open(path):
        m = RE_UNIFORM.match(line)
        if not m:
            continue
        if "MISMATCH" in m.group(3):
            raise SystemExit(f"{path}: {line.strip()} -- a sweep with a mismatch is not a basis for anything")
        T, W = int(m.group(1)), int(m.group(2))
        v = [int(x) for x in m.groups()[3:]]
        out[("unpack", T, W)] = (v[0], v[1:6])
        out[("pack", T, W)] = (v[6], v[7:12])
    return out


def parse_chain(path):
    out = {}
    for line in open(path):
        m = RE_CHAIN.match(line)
        if not m:
            continue
        if m.group(4):
            raise SystemExit(f"{path}: {line.strip()} -- mismatch")
        T, W, name = int(m.group(1)), int(m.group(2)), m.group(3)
        v = [int(x) for x in m.groups()[4:]]
        if name in CHAIN_W_OPS:
            out[(CHAIN_W_OPS[name], T, W)] = (v[0], v[1:6])
        elif name in CHAIN_T_OPS:
            out[(name, T, T)] = (v[0], v[1:6])
    return out


def decide(samples, margin):
    """samples: one (cc, [wpb...]) per box -> waves (0 = cell-column)."""
    if any(cc <= 0 for cc, _ in samples):
        return WAVES[0]
    best_k, best_gm = 0, -1.0
    for k in range(len(WAVES)):
        gm = math.exp(sum(math.log(max(w[k], 1) / cc) for cc, w in samples) / len(samples))
        if gm > best_gm:
            best_k, best_gm = k, gm
    if all(cc > (1.0 + margin) * w[best_k] for cc, w in samples):
        return 0
    return WAVES[best_k]


def decide_two(one, two, margin):
    """one / two: per box (cc, [wpb...]) of the one-block and the two-block form -> waves, TWO_BLOCKS + waves, or 0 (cell-column).
    The two-block form is taken only where its best geometric-mean rate leads the one-block form's best by more than the margin."""
    def best(samples, first=0):
        bk, bg = first, -1.0
        for k in range(first, len(WAVES)):
            gm = math.exp(sum(math.log(max(w[k], 1) / max(cc, 1)) for cc, w in samples) / len(samples))
            if gm > bg:
                bk, bg = k, gm
        return bk, bg
    k1, g1 = best(one)
    # the two-block form is not offered at 3 workgroups per CU: it wins some 45-GB sweep cells there by 2-3 % and falls off a cliff
    # on 8-GB columns (u32 W=24: 0.668 of the peak against 0.80 at 4+; profiles/exp_two_blocks_r05.txt)
    k2, g2 = best(two, first=1)
    use_two = g2 > (1.0 + margin) * g1
    chosen, k = (two, k2) if use_two else (one, k1)
    if all(cc > 0 for cc, _ in chosen) and all(cc > (1.0 + margin) * w[k] for cc, w in chosen):
        return 0
    return (TWO_BLOCKS if use_two else 0) + WAVES[k]


def build(uniform_files, chain_files, margin):
    boxes_u = [parse_uniform(f) for f in uniform_files]
    boxes_c = [parse_chain(f) for f in chain_files]
    table = {}
    for op in UNIFORM_OPS:
        for T in TYPES:
            row = []
            for W in range(T + 1):
                samples = [b[(op, T, W)] for b in boxes_u if (op, T, W) in b]
                if len(samples) != len(boxes_u) or not samples:
                    raise SystemExit(f"uniform sweep lacks {op} u{T} W={W} on some box")
                if op == "pack" and W == 0:
                    row.append(WAVES[0])          # macros.rs:52-53: W == 0 writes nothing -- no kernel runs either way
                else:
                    row.append(decide(samples, margin))
            table[(op, T)] = row
    for op in CHAIN_W_OPS.values():
        if op == "undelta_pack_2b":
            continue                                  # not a row of its own: folded into undelta_pack below
        if op in FOR_FOLLOWS and not any(k[0] == op for b in boxes_c for k in b):
            for T in TYPES:
                table[(op, T)] = list(table[(FOR_FOLLOWS[op], T)])
            continue
        for T in TYPES:
            row = []
            for W in range(T + 1):
                samples = [b[(op, T, W)] for b in boxes_c if (op, T, W) in b]
                if len(samples) != len(boxes_c) or not samples:
                    raise SystemExit(f"chain sweep lacks {op} u{T} W={W} on some box (run tools/abchain.py --all)")
                two = [b[("undelta_pack_2b", T, W)] for b in boxes_c if ("undelta_pack_2b", T, W) in b] if op == "undelta_pack" else []
                if two and len(two) == len(boxes_c) and W > 0:
                    row.append(decide_two(samples, two, margin))
                else:
                    row.append(WAVES[0] if (op in ("transpose_delta_pack", "for_pack") and W == 0) else decide(samples, margin))
            table[(op, T)] = row
    for op in CHAIN_T_OPS:
        for T in TYPES:
            samples = [b[(op, T, T)] for b in boxes_c if (op, T, T) in b]
            if len(samples) != len(boxes_c) or not samples:
                raise SystemExit(f"chain sweep lacks {op} u{T}")
            table[(op, T)] = [decide(samples, margin)]
    return table


def render(table, uniform_files, chain_files, margin):
    rel = lambda f: os.path.relpath(os.path.abspath(f), ROOT)
    lines = [
        "// fl_dispatch_table.inc -- GENERATED by tools/make_dispatch.py; do not edit by hand.",
        "// Inputs (same-buffer A/B sweeps, GB/s; cell-column vs wave-per-block at 3/4/5/6/8 waves per SIMD):",
    ]
    lines += [f"//   uniform: {rel(f)}" for f in uniform_files]
    lines += [f"//   chain:   {rel(f)}" for f in chain_files]
    lines += [
        f"// Rule: k* = occupancy with the best geometric-mean rate over the boxes; the cell-column kernel (entry 0) is kept only",
        f"// where it leads wave-per-block@k* by more than {margin * 100:.0f} % on EVERY box; otherwise the entry is k* (waves per SIMD).",
        "// Index = width W (0..T); the per-type ops have one entry.  UNDELTA_PACK of u32 / u64: an entry 10 + k = the two-blocks-per-wavefront",
        f"// form at k >= 4 waves per SIMD, taken where its best geometric-mean rate leads the one-block form's best by more than {margin * 100:.0f} %.",
        "namespace fl { namespace dispatch_table {",
    ]
    order = list(UNIFORM_OPS) + [o for o in CHAIN_W_OPS.values() if o != "undelta_pack_2b"] + list(CHAIN_T_OPS)
    for op in order:
        for T in TYPES:
            row = table[(op, T)]
            lines.append(f"constexpr unsigned char {op.upper()}_U{T}[{len(row)}] = {{{', '.join(str(x) for x in row)}}};")
    lines += ["}}  // namespace fl::dispatch_table", ""]
    return "\n".join(lines)


def summary(table):
    out = []
    for (op, T), row in sorted(table.items()):
        cc = [w for w, x in enumerate(row) if x == 0]
        out.append(f"{op:26s} u{T:<2d}: cell-column for {len(cc):2d} of {len(row):2d} widths" + (f"  {cc}" if cc and len(row) > 1 else ""))
    return "\n".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--uniform", nargs="+", required=True)
    ap.add_argument("--chain", nargs="+", required=True)
    ap.add_argument("--margin", type=float, default=0.02)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    table = build(a.uniform, a.chain, a.margin)
    text = render(table, a.uniform, a.chain, a.margin)
    if a.check:
        if not os.path.exists(OUT) or open(OUT).read() != text:
            sys.exit(f"{os.path.relpath(OUT, ROOT)} is not what tools/make_dispatch.py generates from these sweeps")
        print("dispatch table is up to date")
        return
    open(OUT, "w").write(text)
    print(summary(table))
    print("wrote", os.path.relpath(OUT, ROOT))


if __name__ == "__main__":
    main()
