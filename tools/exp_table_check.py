#!/usr/bin/env python3
"""Is the generated dispatch table (plus its post-rules) a loser anywhere once the buffers live in a constructed pair?  For every (T, W)
and op in {unpack, pack, undelta_pack}: fl_internal_selftune_check (the table's choice against the cell-column kernel where it is built
and the wave-per-block kernel at 3 / 4 / 5 / 6 / 8 waves per SIMD, same buffers) on a ~12-GB constructed pair.
    python tools/exp_table_check.py [--types u32,u64] [--gb 12]      (one process per type: tools/gpu/table_check.sh)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastlanes_amd as fl
from fastlanes_amd import placement as pl

lib = fl.load(); dev = torch.device("cuda:0")
BITS = {"u8": 8, "u16": 16, "u32": 32, "u64": 64}
types = sys.argv[sys.argv.index("--types") + 1].split(",") if "--types" in sys.argv else list(BITS)
GB = float(sys.argv[sys.argv.index("--gb") + 1]) if "--gb" in sys.argv else 12.0
lib.fl_internal_selftune_check.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                           ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]
lib.fl_internal_pair_chunk_cache(96)
losers = []
for ty in types:
    T = BITS[ty]
    for W in range(1, T + 1):
        for op, name in ((0, "unpack"), (1, "pack"), (2, "undelta_pack")):
            n = int(GB * 1e9 / (128 * (T + W) + (128 if op == 2 else 0)))
            ib, ob = (n * 128 * T, n * 128 * W) if op == 1 else (n * 128 * W, n * 128 * T)
            pair = pl.ColumnPair(ib, ob, dev, aux_bytes=n * 128 if op == 2 else 0, layout="interleaved")
            assert lib.fl_fill_random(pair.input.data_ptr(), ib & ~7, 5, None) == 0
            if op == 2:
                assert lib.fl_fill_random(pair.aux.data_ptr(), (n * 128) & ~7, 6, None) == 0
            t, o, p = ctypes.c_float(), ctypes.c_float(), ctypes.c_int()
            rc = lib.fl_internal_selftune_check(op, T, W, pair.input.data_ptr(), pair.aux.data_ptr() if op == 2 else None, pair.output.data_ptr(), n, None,
                                                ctypes.byref(t), ctypes.byref(o), ctypes.byref(p))
            pair.free()
            if rc != 0 or t.value <= 0 or o.value <= 0:
                print(f"{name:13s} {ty:3s} W={W:<2d} rc={rc}", flush=True)
                continue
            behind = (t.value / o.value - 1) * 100
            alt = "cell-column" if p.value == 1 else f"wave-per-block at {p.value >> 8} waves"
            print(f"{name:13s} {ty:3s} W={W:<2d} table {t.value:7.4f} ms  best alternative {o.value:7.4f} ms ({alt})  table behind by {behind:+5.1f} %", flush=True)
            if behind > 2.0:
                losers.append((behind, name, ty, W, alt))
print(f"# rows where the table's choice is more than 2 % behind the best alternative: {len(losers)}")
for b, name, ty, W, alt in sorted(losers, reverse=True):
    print(f"#   {name} {ty} W={W}: {b:+.1f} % behind {alt}")
