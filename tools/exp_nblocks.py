import sys
sys.path.insert(0, '/root/repo')
import torch
import fastlanes_amd as fl
from bench import rand_u8
dev = torch.device('cuda', 0)
nmax = 10_500_000
pk = rand_u8(nmax * 896, 1, dev).view(torch.uint32)
out = torch.empty(nmax * 1024, dtype=torch.uint32, device=dev)
def rate(n):
    f = lambda: fl.BitPacking.unpack(7, pk[:n * 224], output=out[:n * 1024])
    f(); torch.cuda.synchronize()
    ms = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); b.synchronize(); ms.append(a.elapsed_time(b))
    return n * 4992 / sorted(ms)[3] / 1e6
for n in (1 << 22, (1 << 22) + 1024, (1 << 22) + 40000, 5_000_000, 1 << 23, (1 << 23) + 1024, (1 << 23) + 4096, (1 << 23) + 65536, (1 << 23) + 300000, 8_500_000, 9_000_000, 10_000_000, 10_000_000 + 12345, 10_485_760):
    print(f"n_blocks {n:>9d} ({n / (1 << 20):7.3f} Mi)  {rate(n):7.0f} GB/s", flush=True)
