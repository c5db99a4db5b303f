"""Scratch experiment 2 (GPU box): same kernel, same process, buffers from torch's allocator vs raw hipMalloc."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastlanes_amd as fl
from bench import Workload

dev = torch.device("cuda", 0)
n = 9_765_625
w = Workload("u32_mixed_unpack", n, 0, 0, dev)
lib = fl.load()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]

def raw(nbytes):
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), nbytes) == 0
    return p.value

def run(label, call, reps=12):
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); call(); b.record()
        ms.append((a, b))
    torch.cuda.synchronize()
    t = sorted(x.elapsed_time(y) for x, y in ms)
    print(f"{label:64s} med {t[len(t)//2]:8.4f} ms  min {t[0]:8.4f}  -> {w.bytes / t[len(t)//2] / 1e6:8.1f} GB/s", flush=True)

W, O, S, D = w.widths.data_ptr(), w.offsets.data_ptr(), w.src.data_ptr(), w.dst.data_ptr()
f = lib.fl_u32_unpack_widths
run("torch buffers, python mirror", w.step)
run("torch buffers, raw ctypes call", lambda: f(W, O, S, D, n, None, None))
rs, rd = raw(w.src.numel() * 4 + (64 << 20)), raw(n * 4096 + (64 << 20))
hip.hipMemcpy(rs, S, w.src.numel() * 4, 3)
run("hipMalloc packed + out, torch widths/offsets", lambda: f(W, O, rs, rd, n, None, None))
rw, ro = raw(n + 64), raw(n * 8 + 64)
hip.hipMemcpy(rw, W, n, 3); hip.hipMemcpy(ro, O, n * 8, 3)
run("hipMalloc everything", lambda: f(rw, ro, rs, rd, n, None, None))
run("hipMalloc packed, torch out", lambda: f(W, O, rs, D, n, None, None))
run("torch packed, hipMalloc out", lambda: f(W, O, S, rd, n, None, None))
print("torch ptrs: src %x dst %x   raw: src %x dst %x" % (S, D, rs, rd))
