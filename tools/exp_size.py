"""Scratch (GPU box): does the best occupancy of the wave-per-block unpack depend on the column size?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastlanes_amd as fl
from bench import rand_u8
lib = fl.load()
dev = torch.device("cuda", 0)
for ty, tdt, T, W in (("u64", torch.uint64, 64, 17), ("u64", torch.uint64, 64, 24), ("u32", torch.uint32, 32, 12), ("u32", torch.uint32, 32, 30)):
    esz = T // 8
    for n in (1_000_000, 2_500_000, 5_000_000, 10_000_000):
        pk = rand_u8(n * 128 * W, 2, dev).view(tdt)
        un = torch.empty(n * 1024, dtype=tdt, device=dev)
        nbytes = n * (128 * W + 1024 * esz)
        res = {}
        pols = [1] + [2 + 256 * w for w in (3, 4, 5, 6)]
        for _ in range(3):
            for p in pols:
                lib.fl_set_kernel_policy(p)
                fl.BitPacking.unpack(W, pk, output=un); torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); fl.BitPacking.unpack(W, pk, output=un); b.record(); torch.cuda.synchronize()
                res.setdefault(p, []).append(a.elapsed_time(b))
        g = [nbytes / sorted(res[p])[1] / 1e6 for p in pols]
        print(f"{ty} W={W:<2d} unpack n={n:>9d} ({nbytes / 1e9:6.1f} GB) | cc {g[0]:6.0f} | wpb 3/4/5/6 waves " + " ".join(f"{x:6.0f}" for x in g[1:]), flush=True)
        del pk, un
    lib.fl_set_kernel_policy(0)
