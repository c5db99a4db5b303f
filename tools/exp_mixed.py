"""Scratch experiment (GPU box): why does bench.py's mixed leg read ~5 % below tools/abmixed on the same box?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastlanes_amd as fl
from bench import Workload, rand_u8

dev = torch.device("cuda", 0)
n = 9_765_625
w = Workload("u32_mixed_unpack", n, 0, 0, dev)

def run(label, step, between=None, sync_each=False, reps=12):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        if between:
            between()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); step(); b.record()
        if sync_each:
            torch.cuda.synchronize()
        ms.append((a, b))
    torch.cuda.synchronize()
    t = sorted(x.elapsed_time(y) for x, y in ms)
    print(f"{label:60s} med {t[len(t)//2]:8.4f} ms  min {t[0]:8.4f}  -> {w.bytes / t[len(t)//2] / 1e6:8.1f} GB/s", flush=True)

run("back-to-back (bench.py's method)", w.step)
run("sync after each launch", w.step, sync_each=True)
other_src = rand_u8(2_000_000 * 896, 5, dev).view(torch.uint32)
other_dst = torch.empty(2_000_000 * 1024, dtype=torch.uint32, device=dev)
run("another kernel (uniform unpack, 10 GB) between launches", w.step, between=lambda: fl.BitPacking.unpack(7, other_src, output=other_dst))
# offsets as given vs recomputed; widths int pattern the same -- now different allocation order: dst first
del w.dst
torch.cuda.empty_cache()
w.dst = torch.empty(n * 1024, dtype=torch.uint32, device=dev)
run("dst re-allocated", w.step)
# offset the packed column by 1 MiB + 128 inside a bigger allocation
big = torch.empty(w.src.numel() + (1 << 20), dtype=torch.uint32, device=dev)
sh = big[(1 << 18) + 32:(1 << 18) + 32 + w.src.numel()]
sh.copy_(w.src)
old = w.src
w.src = sh
run("packed column shifted by 1 MiB + 128 B", w.step)
w.src = old
# the round-1 style plan object on the same buffers
import numpy as np
plan = fl.MixedWidthPlan("u32", (1 + np.arange(n) % 32).astype(np.uint8))
run("MixedWidthPlan.unpack (same kernel via the plan)", lambda: plan.unpack(w.src, output=w.dst))
run("back-to-back again", w.step)
