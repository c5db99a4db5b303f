#!/usr/bin/env python3
"""Experiment: does the on-device FoR encoder (block_min_max -> for_widths -> widths_to_offsets -> for_pack_widths, two passes over the
values) get its second pass out of the 256-MiB Infinity Cache when the column is encoded CHUNK by chunk?  The four launches of every
chunk are captured into one HIP graph (no per-launch host time); every chunk packs into its own worst-case region (timing only).
    python tools/exp_chunked_encoder.py [--gb 16] [--reps 5]
Prints ms and T ints/s for the whole column at once and for chunks of 256 / 128 / 64 / 32 / 16 MiB of values."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlanes_amd as fl  # noqa: E402

lib = fl.load()
dev = torch.device("cuda:0")
ap = argparse.ArgumentParser()
ap.add_argument("--gb", type=float, default=16.0)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()

T, esz, ty = 32, 4, "u32"
n = int(args.gb * 1e9 / 4096)
g = torch.Generator(device=dev)
g.manual_seed(9)
widths = torch.randint(1, T, (n,), dtype=torch.int64, device=dev, generator=g).to(torch.uint8)
offsets, total = fl.widths_to_offsets(ty, widths)
col = torch.empty(int(total.item()) // esz, dtype=torch.uint32, device=dev)
assert lib.fl_fill_random(col.data_ptr(), col.numel() * 4 & ~7, 1, None) == 0
refs = torch.randint(0, 1 << 31, (n,), dtype=torch.int64, device=dev, generator=g).to(torch.int32).view(torch.uint32)
un = fl.unfor_pack_widths(widths, offsets, col, refs)           # values with per-block ranges: what an encoder is given
del col
mins = torch.empty(n, dtype=torch.uint32, device=dev)
maxs = torch.empty(n, dtype=torch.uint32, device=dev)
w2 = torch.empty(n, dtype=torch.uint8, device=dev)
off2 = torch.empty(n, dtype=torch.int64, device=dev)
tot = torch.zeros(1, dtype=torch.int64, device=dev)
err = torch.zeros(1, dtype=torch.int32, device=dev)
packed = torch.empty(n * 1024, dtype=torch.uint32, device=dev)   # worst case: every chunk's region starts at its unpacked offset
P = ctypes.c_void_p


def encode(chunk_blocks, stream):
    for first in range(0, n, chunk_blocks):
        nb = min(chunk_blocks, n - first)
        v = un.data_ptr() + first * 4096
        lo, hi = mins.data_ptr() + first * 4, maxs.data_ptr() + first * 4
        w, o = w2.data_ptr() + first, off2.data_ptr() + first * 8
        pk = packed.data_ptr() + first * 4096
        assert lib.fl_u32_block_min_max(P(v), nb, P(lo), P(hi), stream) == 0
        assert lib.fl_u32_for_widths(P(lo), P(hi), nb, P(w), stream) == 0
        assert lib.fl_widths_to_offsets(32, P(w), nb, P(o), P(tot.data_ptr()), P(err.data_ptr()), stream) == 0
        assert lib.fl_u32_for_pack_widths(P(w), P(o), P(v), P(lo), 1, P(pk), nb * 4096, nb, P(err.data_ptr()), stream) == 0


print(f"# {lib.fl_version().decode()}\n# u32, {n} blocks ({n * 4096 / 1e9:.1f} GB of values), widths 1..31; median of {args.reps}; one HIP graph per row", flush=True)
for label, chunk in (("whole column", n), ("256 MiB", 65536), ("128 MiB", 32768), ("64 MiB", 16384), ("32 MiB", 8192), ("16 MiB", 4096)):
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        encode(chunk, P(torch.cuda.current_stream().cuda_stream))
    graph.replay()
    torch.cuda.synchronize()
    ms = []
    for _ in range(args.reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); graph.replay(); b.record(); b.synchronize()
        ms.append(a.elapsed_time(b))
    med = sorted(ms)[len(ms) // 2]
    assert int(err.item()) == 0
    print(f"chunks of {label:13s} ({(n + chunk - 1) // chunk:5d} x 4 launches): {med:8.3f} ms  {n * 1024 / med / 1e9:6.3f} T ints/s", flush=True)
    del graph
# correctness of the chunked form (last row): decode every chunk and compare
ok = True
for first in range(0, n, 4096 * 64):
    nb = min(4096, n - first)
    back = fl.unfor_pack_widths(w2[first:first + nb], off2[first:first + nb], packed[first * 1024:(first + nb) * 1024], mins[first:first + nb])
    ok &= bool(torch.equal(back.view(torch.int32), un[first * 1024:(first + nb) * 1024].view(torch.int32)))
print("round trip of sampled chunks:", "ok" if ok else "MISMATCH")
