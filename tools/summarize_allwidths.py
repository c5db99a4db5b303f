#!/usr/bin/env python3
"""Recomputes the per-(op, type) summary of a tools/sweep.py --cases allwidths output made of several processes' rows (one process per
element type, the wide types in two width ranges: a constructed pair's address ranges are never re-used within a process).
    python tools/summarize_allwidths.py <rows file>      (prints the rows, comments and a fresh summary; drops the per-process summaries)"""
import re
import sys

rows, keep = [], []
for l in open(sys.argv[1]):
    if l.startswith("# ----") or re.match(r"# \w+\s+u\d+\s+min ", l):
        continue
    keep.append(l.rstrip("\n"))
    m = re.match(r"(\w+)\s+(u\d+)\s+W=(\d+)\s+n=\s*\d+\s+[\d.]+ ms\s+[\d.]+ GB/s ([\d.]+)(?:.*?-> ([\d.]+) of it)?", l)
    if m:
        rows.append((m.group(1), m.group(2), int(m.group(3)), float(m.group(4)), float(m.group(5)) if m.group(5) else None))
print("\n".join(keep))
print("# ---- summary: fraction of the 8 TB/s peak per (op, type) over all widths 1..T: min (at W) / median / max (at W)")
ops = list(dict.fromkeys(r[0] for r in rows))
for op in ops:
    for ty in ("u8", "u16", "u32", "u64"):
        rs = sorted((r[3], r[2]) for r in rows if r[0] == op and r[1] == ty)
        ob = sorted((r[4], r[2]) for r in rows if r[0] == op and r[1] == ty and r[4])
        if not rs:
            continue
        print(f"# {op:24s} {ty:4s} min {rs[0][0]:.3f} (W={rs[0][1]:<2d})  median {rs[len(rs) // 2][0]:.3f}  max {rs[-1][0]:.3f} (W={rs[-1][1]:<2d})" +
              (f"   | of the bare stream of the same bytes on the same buffers: min {ob[0][0]:.3f} (W={ob[0][1]:<2d})  median {ob[len(ob) // 2][0]:.3f}" if ob else ""))
worst = sorted(rows, key=lambda r: r[3])[:8]
print("# ---- the eight slowest (op, T, W): " + "; ".join(f"{r[0]} {r[1]} W={r[2]} {r[3]:.3f}" + (f" ({r[4]:.2f} of its bare stream)" if r[4] else "") for r in worst))
