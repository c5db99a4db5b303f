#!/usr/bin/env python3
"""A/B on the GPU box: the same entry points of several BUILDS of the library on the same buffers, interleaved -- e.g. builds
before / after a kernel change, or with an experiment macro set differently (round 3: waves-per-SIMD caps of the fused
consumers, occupancy of the narrow types' cell-column kernels).
    python tools/ablibs.py <rounds> <ops> <cases> lib_a.so lib_b.so ...
      ops    comma list of: unpack pack undelta_pack undelta_pack_untranspose undelta compare sums   (undelta ignores the width)
      cases  comma list of type:width, e.g. u8:3,u8:6,u16:3   (or "consumers" = round 3's consumer sweep)
      ops    also: unpack_widths (the width of the case is ignored: width[b] = 1 + b mod T, BASELINE config 5's ramp),
             undelta_pack_widths undelta_pack_untranspose_widths transpose_delta_pack_widths (the same ramp), transpose_delta_pack,
             transpose, untranspose
    FL_AB_BLOCKS=<n> in the environment: that many blocks per case instead of ~8 GB of traffic (BASELINE sizes: 10000000).
GB/s of algorithmic bytes, median and best."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import rand_u8  # noqa: E402

ROUNDS = int(sys.argv[1])
OPS = sys.argv[2].split(",")
if sys.argv[3] == "consumers":
    cases = [("u32", w) for w in (2, 4, 7, 10, 12, 16, 20, 24, 28, 32)] + [("u64", w) for w in (4, 8, 12, 17, 24, 40, 56)] + \
            [("u16", w) for w in (3, 6, 9, 12, 16)] + [("u8", w) for w in (3, 6, 8)]
else:
    cases = [(c.split(":")[0], int(c.split(":")[1])) for c in sys.argv[3].split(",")]
libs = [(os.path.basename(p), ctypes.CDLL(os.path.abspath(p))) for p in sys.argv[4:]]
dev = torch.device("cuda", 0)
TD = {"u8": (torch.uint8, 8), "u16": (torch.uint16, 16), "u32": (torch.uint32, 32), "u64": (torch.uint64, 64)}
CT = {"u8": ctypes.c_uint8, "u16": ctypes.c_uint16, "u32": ctypes.c_uint32, "u64": ctypes.c_uint64}
P, Z, U = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint
print("GB/s, median of %d; columns: %s" % (ROUNDS, "  ".join(n for n, _ in libs)))
for ty, W in cases:
    tdt, T = TD[ty]
    esz = T // 8
    for op in OPS:
        bpb = {"compare": 128 * W + 128, "sums": 128 * W + 8, "undelta_pack": 128 * W + 128 + 128 * T,
               "undelta_pack_untranspose": 128 * W + 128 + 128 * T, "undelta": 2 * 128 * T + 128}.get(op, 128 * W + 128 * T)
        n = int(os.environ.get("FL_AB_BLOCKS", 0)) or int(8e9 / bpb)
        widths = offsets = None
        if op.endswith("_widths"):
            widths = (1 + torch.arange(n, dtype=torch.int64, device=dev) % T).to(torch.uint8)
            offsets = torch.cumsum(widths.to(torch.int64) * 128, 0) - widths.to(torch.int64) * 128
            pbytes = int(offsets[-1].item()) + 128 * int(widths[-1].item())
            bpb = (pbytes + n * 128 * T) / n
        if op.endswith("_widths") and "delta" in op:
            bpb += 128
        if op in ("transpose", "untranspose"):
            bpb = 2 * 128 * T
        if op == "transpose_delta_pack":
            bpb = 128 * W + 128 + 128 * T
        encode = op in ("pack", "undelta", "transpose", "untranspose", "transpose_delta_pack", "transpose_delta_pack_widths", "pack_widths")
        pk = rand_u8(pbytes if op.endswith("_widths") else n * 128 * W, 2, dev).view(tdt)
        un = rand_u8(n * 128 * T, 3, dev).view(tdt) if encode else None
        bases = rand_u8(n * 128, 4, dev).view(tdt) if "delta" in op else None
        out_bytes = {"compare": n * 128, "sums": n * 8, "pack": n * 128 * W, "transpose_delta_pack": n * 128 * W}.get(op, n * 128 * T)
        if op in ("transpose_delta_pack_widths", "pack_widths"):
            out_bytes = pbytes
        out = torch.empty(max(out_bytes, 16) // 4, dtype=torch.int32, device=dev)
        fns = []
        for _, lib in libs:
            if op == "compare":
                f = getattr(lib, f"fl_{ty}_unpack_compare"); f.argtypes = [U, P, ctypes.c_int, CT[ty], Z, P, P]
                fns.append(lambda f=f: f(W, pk.data_ptr(), 2, (1 << W) // 2, n, out.data_ptr(), None))
            elif op == "sums":
                f = getattr(lib, f"fl_{ty}_unpack_block_sums"); f.argtypes = [U, P, Z, P, P]
                fns.append(lambda f=f: f(W, pk.data_ptr(), n, out.data_ptr(), None))
            elif op == "unpack_widths":
                f = getattr(lib, f"fl_{ty}_unpack_widths"); f.argtypes = [P, P, P, Z, P, Z, P, P]
                fns.append(lambda f=f: f(widths.data_ptr(), offsets.data_ptr(), pk.data_ptr(), pbytes, out.data_ptr(), n, None, None))
            elif op in ("undelta_pack_widths", "undelta_pack_untranspose_widths"):
                f = getattr(lib, f"fl_{ty}_{op}"); f.argtypes = [P, P, P, Z, P, P, Z, P, P]
                fns.append(lambda f=f: f(widths.data_ptr(), offsets.data_ptr(), pk.data_ptr(), pbytes, bases.data_ptr(), out.data_ptr(), n, None, None))
            elif op == "pack_widths":
                f = getattr(lib, f"fl_{ty}_pack_widths"); f.argtypes = [P, P, P, P, Z, Z, P, P]
                fns.append(lambda f=f: f(widths.data_ptr(), offsets.data_ptr(), un.data_ptr(), out.data_ptr(), pbytes, n, None, None))
            elif op == "transpose_delta_pack_widths":
                f = getattr(lib, f"fl_{ty}_{op}"); f.argtypes = [P, P, P, P, P, Z, Z, P, P]
                fns.append(lambda f=f: f(widths.data_ptr(), offsets.data_ptr(), un.data_ptr(), bases.data_ptr(), out.data_ptr(), pbytes, n, None, None))
            elif op == "transpose_delta_pack":
                f = getattr(lib, f"fl_{ty}_{op}"); f.argtypes = [U, P, P, P, Z, P]
                fns.append(lambda f=f: f(W, un.data_ptr(), bases.data_ptr(), out.data_ptr(), n, None))
            elif op in ("transpose", "untranspose"):
                f = getattr(lib, f"fl_{ty}_{op}"); f.argtypes = [P, P, Z, P]
                fns.append(lambda f=f: f(un.data_ptr(), out.data_ptr(), n, None))
            elif op == "unpack":
                f = getattr(lib, f"fl_{ty}_unpack"); f.argtypes = [U, P, P, Z, P]
                fns.append(lambda f=f: f(W, pk.data_ptr(), out.data_ptr(), n, None))
            elif op == "pack":
                f = getattr(lib, f"fl_{ty}_pack"); f.argtypes = [U, P, P, Z, P]
                fns.append(lambda f=f: f(W, un.data_ptr(), out.data_ptr(), n, None))
            elif op == "undelta":
                f = getattr(lib, f"fl_{ty}_undelta"); f.argtypes = [P, P, P, Z, P]
                fns.append(lambda f=f: f(un.data_ptr(), bases.data_ptr(), out.data_ptr(), n, None))
            else:
                f = getattr(lib, f"fl_{ty}_{op}"); f.argtypes = [U, P, P, P, Z, P]
                fns.append(lambda f=f: f(W, pk.data_ptr(), bases.data_ptr(), out.data_ptr(), n, None))
        ref, same = None, True
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
        best = [n * bpb / min(m) / 1e6 for m in ms]
        delta = "" if len(g) != 2 else f" | {(g[1] / g[0] - 1) * 100:+5.1f} %"
        print(f"{ty:3s} W={W:<2d} {op:24s} n={n:<9d}{'' if same else ' MISMATCH'} | " + " ".join(f"{x:6.0f}" for x in g) +
              " | best " + " ".join(f"{x:6.0f}" for x in best) + delta, flush=True)
        del pk, out, ref, un, bases, widths, offsets
        torch.cuda.empty_cache()
