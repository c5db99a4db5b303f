import os, sys, ctypes
sys.path.insert(0, '/root/repo')
import torch, numpy as np
import fastlanes_amd as fl
from bench import rand_u8
lib = fl.load()
dev = torch.device('cuda', 0)
TD = {"u32": (torch.uint32, 32), "u64": (torch.uint64, 64)}
print("compare GB/s median of 5: cc | wave @3 4 5 6 8")
for ty, ws in (("u32", (2, 4, 7, 10, 12, 16, 20, 24, 28, 31, 32)), ("u64", (4, 8, 12, 17, 24, 40, 56, 64))):
    tdt, T = TD[ty]
    for W in ws:
        bpb = 128 * W + 128
        n = int(8e9 / bpb)
        pk = rand_u8(n * 128 * W, 2, dev).view(tdt)
        f = lambda: fl.BitPacking.unpack_compare(W, pk, "<", (1 << W) // 2, n_blocks=n)
        lib.fl_internal_set_kernel_policy(1); ref = f().clone()
        res = {}
        pols = [1] + [2 + 256 * w for w in (3, 4, 5, 6, 8)]
        same = True
        for p in pols[1:]:
            lib.fl_internal_set_kernel_policy(p)
            same = same and torch.equal(ref, f())
        for _ in range(5):
            for p in pols:
                lib.fl_internal_set_kernel_policy(p)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); f(); b.record(); b.synchronize()
                res.setdefault(p, []).append(a.elapsed_time(b))
        g = [n * bpb / sorted(res[p])[2] / 1e6 for p in pols]
        print(f"{ty} W={W:<2d} {'' if same else 'MISMATCH '}| {g[0]:6.0f} | " + " ".join(f"{x:6.0f}" for x in g[1:]), flush=True)
        lib.fl_internal_set_kernel_policy(0)
        del pk, ref
