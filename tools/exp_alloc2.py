"""Scratch (GPU box): map the speed of the headline kernel over many 4 GiB output allocations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastlanes_amd as fl
from bench import rand_u8
dev = torch.device("cuda", 0)
n = 1 << 20                      # 4 GiB of output per buffer
W = 7
src = rand_u8(n * 896, 10, dev).view(torch.uint32)
bufs = []
while len(bufs) < 60 and torch.cuda.mem_get_info()[0] > (10 << 30):
    bufs.append(torch.empty(n * 1024, dtype=torch.uint32, device=dev))
def t(dst, reps=6):
    for _ in range(2):
        fl.BitPacking.unpack(W, src, output=dst)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fl.BitPacking.unpack(W, src, output=dst); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return n * 4992 / sorted(ts)[len(ts) // 2] / 1e6
res = [t(b) for b in bufs]
print(len(bufs), "buffers of 4 GiB; GB/s per buffer in allocation order:")
for k in range(0, len(res), 10):
    print(" ".join(f"{x:6.0f}" for x in res[k:k + 10]))
print("min %.0f  median %.0f  max %.0f" % (min(res), sorted(res)[len(res) // 2], max(res)))
res2 = [t(b) for b in bufs[:10]]
print("first ten again:", " ".join(f"{x:6.0f}" for x in res2))
