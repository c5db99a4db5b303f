#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: a calibration copy of known size
(torch elementwise copy, 16 B/lane) followed by the headline kernels at
BASELINE.json sizes.  Run under
    rocprofv3 --kernel-trace --pmc FETCH_SIZE   --output-format csv -d <dir> -- python tools/pmc_probe.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE   --output-format csv -d <dir> -- python tools/pmc_probe.py
(separate passes: FETCH_SIZE and WRITE_SIZE do not fit the TCC slots together,
MI355X_MICROARCH.md 'rocprofv3 PMC slots')."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlanes_amd as fl  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(1)


def rnd(nbytes):
    return torch.randint(-2**63, 2**63 - 1, (nbytes // 8,), dtype=torch.int64, device=dev, generator=g)


n = int(os.environ.get("FL_BLOCKS", "10000000"))
# calibration: 8 GiB read + 8 GiB written by a plain copy kernel
a = rnd(8 << 30)
b = torch.empty_like(a)
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
del a, b

pk = rnd(n * 896).view(torch.uint32)
out = torch.empty(n * 1024, dtype=torch.uint32, device=dev)
for _ in range(3):
    fl.BitPacking.unpack(7, pk, output=out)
torch.cuda.synchronize()
back = torch.empty(n * 224, dtype=torch.uint32, device=dev)
for _ in range(3):
    fl.BitPacking.pack(7, out, output=back)
torch.cuda.synchronize()
del pk, out, back

pk = rnd(n * 1536).view(torch.uint32)
bases = rnd(n * 128).view(torch.uint32)
out = torch.empty(n * 1024, dtype=torch.uint32, device=dev)
for _ in range(3):
    fl.Delta.undelta_pack(12, pk, bases, output=out)
torch.cuda.synchronize()
del pk, out, bases

pk = rnd(n * 2176).view(torch.uint64)
out = torch.empty(n * 1024, dtype=torch.uint64, device=dev)
for _ in range(3):
    fl.BitPacking.unpack(17, pk, output=out)
torch.cuda.synchronize()
back = torch.empty(n * 272, dtype=torch.uint64, device=dev)
for _ in range(3):
    fl.BitPacking.pack(17, out, output=back)
torch.cuda.synchronize()
print("pmc_probe done")
