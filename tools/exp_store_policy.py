#!/usr/bin/env python3
"""Round 6 (second half): the cache policy of the streaming stores (buffer-store aux bits; shipped: 18 = sc1 | nt) re-measured with the
buffers in a constructed pair -- builds `make -C fastlanes_amd/csrc STOREAUX=n` (libfastlanes_amd_st<n>.so), the same entry point of every
build on the SAME pair, launches round-robin.  Round 3 had compared the policies across processes in plain memory (+-3 % of placement noise).
    python tools/exp_store_policy.py [0,2,3,16,17,19]"""
import ctypes, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastlanes_amd as fl
from fastlanes_amd import placement as pl

lib0 = fl.load(); dev = torch.device("cuda:0")
auxes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,2,3,16,17,19").split(",")]
here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fastlanes_amd")
libs = [("18 (shipped)", lib0)] + [(str(a), ctypes.CDLL(os.path.join(here, f"libfastlanes_amd_st{a}.so"))) for a in auxes]
P, Z, U = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint
CASES = [("unpack u32 W=7", "fl_u32_unpack", 32, 7, 10_000_000), ("unpack u64 W=17", "fl_u64_unpack", 64, 17, 5_000_000), ("unpack u16 W=3", "fl_u16_unpack", 16, 3, 10_000_000),
         ("transpose u32", "fl_u32_transpose", 32, 32, 5_000_000)]
print("fraction of 8 TB/s, median of 7 round-robin launches; columns: store aux " + " | ".join(n for n, _ in libs))
for name, sym, T, W, n in CASES:
    ib, ob = n * 128 * W, n * 128 * T
    pair = pl.ColumnPair(ib, ob, dev, layout="interleaved")
    assert lib0.fl_fill_random(pair.input.data_ptr(), ib & ~7, 5, None) == 0
    fns = []
    for _, L in libs:
        f = getattr(L, sym)
        if "transpose" in sym:
            f.argtypes = [P, P, Z, P]; fns.append(lambda f=f: f(pair.input.data_ptr(), pair.output.data_ptr(), n, None))
        else:
            f.argtypes = [U, P, P, Z, P]; fns.append(lambda f=f: f(W, pair.input.data_ptr(), pair.output.data_ptr(), n, None))
    ms = [[] for _ in fns]
    for r in range(8):
        for k, f in enumerate(fns):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); assert f() == 0; b.record(); b.synchronize()
            if r: ms[k].append(a.elapsed_time(b))
    print(f"{name:16s} {pair.classes[:40]:40s} " + " | ".join(f"{(ib + ob) / statistics.median(m) / 8e9:.3f}" for m in ms), flush=True)
    pair.free()
