"""Scratch (GPU box): default hipMalloc vs hipExtMallocWithFlags(hipDeviceMallocContiguous) for the 40 GB output / 9 GB input."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastlanes_amd as fl
from bench import rand_u8
lib = fl.load()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
dev = torch.device("cuda", 0)
n, W = 10_000_000, 7
def alloc(nbytes, flags):
    p = ctypes.c_void_p()
    rc = hip.hipMalloc(ctypes.byref(p), nbytes) if flags is None else hip.hipExtMallocWithFlags(ctypes.byref(p), nbytes, flags)
    return p.value if rc == 0 else None
def t(src, dst, reps=6):
    f = lambda: lib.fl_u32_unpack(W, src, dst, n, None)
    for _ in range(2): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return n * 4992 / sorted(ts)[len(ts) // 2] / 1e6
seed = rand_u8(n * 896, 3, dev)
srcs = {"default": alloc(n * 896, None), "contiguous": alloc(n * 896, 4)}
for k, p in srcs.items():
    print("src", k, hex(p) if p else "ALLOC FAILED")
    if p: hip.hipMemcpy(p, seed.data_ptr(), n * 896, 3)
hip.hipFree.argtypes = [ctypes.c_void_p]
n = 5_000_000          # 20 GB outputs: 12 samples fit
for rnd in range(2):
    order = [("default", None), ("contiguous", 4)] * 6 if rnd == 0 else [("contiguous", 4), ("default", None)] * 6
    got = []
    row = {"default": [], "contiguous": []}
    for name, fl_ in order:
        d = alloc(n * 4096, fl_)
        if d is None:
            print("dst", name, "ALLOC FAILED"); continue
        got.append(d)
        row[name].append(t(srcs["default"], d))
    for k, v in row.items():
        print(f"round {rnd} {k:10s}: " + " ".join(f"{x:6.0f}" for x in v) + f"   mean {sum(v) / len(v):6.0f}", flush=True)
    for d in got:
        hip.hipFree(d)
