#!/usr/bin/env python3
"""A/B on the GPU box: mixed-width columns of every element type (seeded-random widths 1..T), unpack_widths and pack_widths at
several blocks-per-wavefront x waves-per-SIMD, same buffers (fl_internal_set_kernel_policy 2 + 256*waves + 65536*bpw).  GB/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import fastlanes_amd as fl  # noqa: E402
from bench import rand_u8  # noqa: E402

lib = fl.load()
dev = torch.device("cuda", 0)
TD = {"u8": (torch.uint8, 8), "u16": (torch.uint16, 16), "u32": (torch.uint32, 32), "u64": (torch.uint64, 64)}
print("mixed widths (seeded random 1..T), ~12 GB columns, GB/s median of 3; rows = blocks per wavefront, columns = waves/SIMD 3 4 6 8")
for ty in ("u8", "u16", "u32", "u64"):
    tdt, T = TD[ty]
    esz = T // 8
    n = (12 << 30) // (128 * (T + 1) // 2 + 128 * T)
    widths = torch.from_numpy(np.random.default_rng(7).integers(1, T + 1, size=n).astype(np.uint8)).to(dev)
    offsets, total = fl.widths_to_offsets(ty, widths)
    pbytes = int(total.item())
    pk = rand_u8(pbytes, 2, dev).view(tdt)
    un = torch.empty(n * 1024, dtype=tdt, device=dev)
    vals = rand_u8(n * 1024 * esz, 3, dev).view(tdt)
    pk2 = torch.empty_like(pk)
    nbytes = pbytes + n * 1024 * esz
    for name, f in (("unpack_widths", lambda: fl.unpack_widths(widths, offsets, pk, output=un, check=False)),
                    ("pack_widths", lambda: fl.pack_widths(widths, offsets, vals, pk2, check=False))):
        shapes = [(1, 0), (2, 0), (4, 0), (8, 0)]
        # + all of a wavefront's blocks requested up front by LDS-DMA (needs bpw * block bytes of LDS per wave)
        shapes += [(b, 1) for b in (2, 4, 8, 16) if 4 * b * 128 * T <= 65536]
        for bpw, prefetch in shapes:
            row = []
            for waves in (3, 4, 6, 8):
                lib.fl_internal_set_kernel_policy(2 + 256 * waves + 65536 * bpw + (prefetch << 24))
                t = []
                for _ in range(3):
                    f(); torch.cuda.synchronize()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(); f(); b.record(); torch.cuda.synchronize()
                    t.append(a.elapsed_time(b))
                row.append(nbytes / sorted(t)[1] / 1e6)
            print(f"{ty:3s} {name:13s} bpw {bpw:2d}{' prefetch' if prefetch else '         '} | " + " ".join(f"{x:6.0f}" for x in row), flush=True)
    lib.fl_internal_set_kernel_policy(0)
    del pk, un, vals, pk2, widths, offsets
