"""Scratch (GPU box): how much does the kernel's speed depend on WHICH allocation the output / input lives in?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastlanes_amd as fl
from bench import rand_u8
dev = torch.device("cuda", 0)
n = 10_000_000
W = 7
def t_unpack(src, dst, reps=8):
    for _ in range(2):
        fl.BitPacking.unpack(W, src, output=dst)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fl.BitPacking.unpack(W, src, output=dst); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return n * 4992 / sorted(ts)[len(ts) // 2] / 1e6
srcs = [rand_u8(n * 896, 10 + k, dev).view(torch.uint32) for k in range(3)]
dsts = [torch.empty(n * 1024, dtype=torch.uint32, device=dev) for k in range(5)]
print("free GB after allocs:", torch.cuda.mem_get_info()[0] / 1e9)
for si, s in enumerate(srcs):
    print(f"src{si} @{s.data_ptr():x}: " + "  ".join(f"dst{di}@{d.data_ptr() >> 30:x}G {t_unpack(s, d):6.0f}" for di, d in enumerate(dsts)), flush=True)
# second pass: is it stable over time?
print("again   src0: " + "  ".join(f"dst{di} {t_unpack(srcs[0], d):6.0f}" for di, d in enumerate(dsts)), flush=True)
# reallocate in a different order
del dsts
torch.cuda.empty_cache()
dsts = [torch.empty(n * 1024, dtype=torch.uint32, device=dev) for k in range(5)]
print("realloc src0: " + "  ".join(f"dst{di}@{d.data_ptr() >> 30:x}G {t_unpack(srcs[0], d):6.0f}" for di, d in enumerate(dsts)), flush=True)
