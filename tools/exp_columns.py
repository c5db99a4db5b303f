#!/usr/bin/env python3
"""Round 5 experiment: the pipelined column-lanes kernels (fl_chain.hpp: k_chain_columns_pipelined / _encode_pipelined) for Delta over
mixed-width u8 columns at several numbers of resident wavefronts per CU -- same buffers, launches interleaved round-robin, outputs
compared.      python tools/exp_columns.py [--types u8,u16] [--gb 8] [--reps 9]
(The rounds of the experiment that compared them with the lockstep kernel and with the non-pipelined forms selected those through
environment knobs that are gone with the variants: profiles/exp_columns_r05.txt holds the results, the history the code.)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlanes_amd as fl  # noqa: E402
from bench import rand_u8  # noqa: E402

ESZ = {"u8": 1, "u16": 2}
TDT = {"u8": torch.uint8, "u16": torch.uint16}
dev = torch.device("cuda:0")
ap = argparse.ArgumentParser()
ap.add_argument("--types", default="u8,u16")
ap.add_argument("--gb", type=float, default=8.0)
ap.add_argument("--reps", type=int, default=9)
ap.add_argument("--variants", default="")
args = ap.parse_args()
VARIANTS = []
VARIANTS += [("shipped", {}), ("8 per CU", {"POLICY": "0x202"}), ("12 per CU", {"POLICY": "0x302"}), ("16 per CU", {"POLICY": "0x402"})]        # (the launch-shape knobs of the round-5 experiments are gone with the variants they selected:
                                     #  profiles/exp_columns_r05.txt holds their results, the history their code)


def select(env):
    for k in ("FL_EXP_LOCKSTEP", "FL_EXP_COLUMNS"):
        os.environ.pop(k, None)
    env = dict(env)
    fl.load().fl_internal_set_kernel_policy(int(env.pop("POLICY", "0"), 0))
    os.environ.update(env)


for ty in args.types.split(","):
    T = ESZ[ty] * 8
    n = int(args.gb * 1e9 / (128 * T * 1.5))
    g = torch.Generator(device=dev)
    g.manual_seed(31 + T)
    widths = torch.randint(1, T, (n,), dtype=torch.int64, device=dev, generator=g).to(torch.uint8)
    offsets, total = fl.widths_to_offsets(ty, widths)
    pbytes = int(total.item())
    col = rand_u8(pbytes, 14, dev).view(TDT[ty])
    bases = rand_u8(n * 128, 15, dev).view(TDT[ty])
    un = torch.empty(n * 1024, dtype=TDT[ty], device=dev)
    nbytes = pbytes + n * 128 * T + n * 128
    for untranspose in (False, True):
        f = lambda: fl.undelta_pack_widths(widths, offsets, col, bases, output=un, check=False, untranspose=untranspose)
        ref, same, ms = None, {}, {}
        for name, env in VARIANTS:
            select(env)
            un.zero_()
            f()
            torch.cuda.synchronize()
            if ref is None:
                ref = un.clone()
            same[name] = torch.equal(ref, un)
            ms[name] = []
        for _ in range(args.reps):
            for name, env in VARIANTS:
                select(env)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); f(); b.record(); b.synchronize()
                ms[name].append(a.elapsed_time(b))
        for name, _ in VARIANTS:
            t = sorted(ms[name])[len(ms[name]) // 2]
            print(f"{ty:3s} undelta_pack{'_untranspose' if untranspose else ''}_widths n={n} {name:34s} {t:8.4f} ms {nbytes / t / 1e6:7.0f} GB/s "
                  f"{nbytes / t / 8e9:.3f}{'' if same[name] else '  MISMATCH'}", flush=True)
    if ty == "u8":
        # the encode side: transpose_delta_pack_widths, the lockstep kernel (FL_EXP_LOCKSTEP_ENCODE=1) against the pipelined column-lanes one
        vals = rand_u8(n * 1024, 16, dev).view(TDT[ty])
        back = torch.empty_like(col)
        f = lambda: fl.transpose_delta_pack_widths(widths, offsets, vals, bases, back, check=False)
        ref, ms, same = None, {}, {}
        enc = [("pipelined column lanes", {})]
        for name, env in enc:
            os.environ.pop("FL_EXP_LOCKSTEP_ENCODE", None)
            os.environ.update(env)
            back.zero_()
            f()
            torch.cuda.synchronize()
            if ref is None:
                ref = back.clone()
            same[name] = torch.equal(ref, back)
            ms[name] = []
        for _ in range(args.reps):
            for name, env in enc:
                os.environ.pop("FL_EXP_LOCKSTEP_ENCODE", None)
                os.environ.update(env)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); f(); b.record(); b.synchronize()
                ms[name].append(a.elapsed_time(b))
        os.environ.pop("FL_EXP_LOCKSTEP_ENCODE", None)
        for name, _ in enc:
            t = sorted(ms[name])[len(ms[name]) // 2]
            print(f"{ty:3s} transpose_delta_pack_widths n={n} {name:34s} {t:8.4f} ms {nbytes / t / 1e6:7.0f} GB/s {nbytes / t / 8e9:.3f}"
                  f"{'' if same[name] else '  MISMATCH'}", flush=True)
        del vals, back
    del col, bases, un, ref
    torch.cuda.empty_cache()
select({})
