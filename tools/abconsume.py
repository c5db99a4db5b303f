#!/usr/bin/env python3
"""A/B on the GPU box: the fused consumers (unpack_compare, unpack_block_sums) of several BUILDS of the library on the same
buffers, interleaved -- e.g. builds before / after a change of the consumer kernels (round 3: waves-per-SIMD caps).
    python tools/abconsume.py [rounds] lib_a.so lib_b.so ...
GB/s of algorithmic bytes (128*W in, 128 B mask / 8 B sum out per block), median."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import rand_u8  # noqa: E402

ROUNDS = int(sys.argv[1])
libs = [(os.path.basename(p), ctypes.CDLL(os.path.abspath(p))) for p in sys.argv[2:]]
dev = torch.device("cuda", 0)
TD = {"u8": (torch.uint8, 8), "u16": (torch.uint16, 16), "u32": (torch.uint32, 32), "u64": (torch.uint64, 64)}
CT = {"u8": ctypes.c_uint8, "u16": ctypes.c_uint16, "u32": ctypes.c_uint32, "u64": ctypes.c_uint64}
P, Z = ctypes.c_void_p, ctypes.c_size_t
cases = [("u32", w) for w in (2, 4, 7, 10, 12, 16, 20, 24, 28, 32)] + [("u64", w) for w in (4, 8, 12, 17, 24, 40, 56)] + \
        [("u16", w) for w in (3, 6, 9, 12, 16)] + [("u8", w) for w in (3, 6, 8)]
print("GB/s, median of %d; columns: %s" % (ROUNDS, "  ".join(n for n, _ in libs)))
for ty, W in cases:
    tdt, T = TD[ty]
    for op in ("compare", "sums"):
        bpb = 128 * W + (128 if op == "compare" else 8)
        n = int(8e9 / bpb)
        pk = rand_u8(n * 128 * W, 2, dev).view(tdt)
        out = torch.empty(n * (32 if op == "compare" else 2), dtype=torch.int32, device=dev)
        fns = []
        for _, lib in libs:
            if op == "compare":
                f = getattr(lib, f"fl_{ty}_unpack_compare")
                f.argtypes = [ctypes.c_uint, P, ctypes.c_int, CT[ty], Z, P, P]
                fns.append(lambda f=f: f(W, pk.data_ptr(), 2, (1 << W) // 2, n, out.data_ptr(), None))
            else:
                f = getattr(lib, f"fl_{ty}_unpack_block_sums")
                f.argtypes = [ctypes.c_uint, P, Z, P, P]
                fns.append(lambda f=f: f(W, pk.data_ptr(), n, out.data_ptr(), None))
        ref = None
        same = True
        for f in fns:
            out.zero_()
            assert f() == 0
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            else:
                same = same and torch.equal(ref, out)
        ms = [[] for _ in fns]
        for _ in range(ROUNDS):
            for k, f in enumerate(fns):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); f(); b.record(); b.synchronize()
                ms[k].append(a.elapsed_time(b))
        g = [n * bpb / sorted(m)[len(m) // 2] / 1e6 for m in ms]
        print(f"{ty:3s} W={W:<2d} {op:7s}{'' if same else ' MISMATCH'} | " + " ".join(f"{x:6.0f}" for x in g), flush=True)
        del pk, out, ref
