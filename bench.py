#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X FastLanes decode path.

One "step" = one pass of `BitPacking::unchecked_unpack` (u32, width 7) over a
column of 10 M 1024-value blocks (BASELINE.json configs[1]), input already
resident in HBM, output materialised to HBM.  With --gpus N every rank decodes
its own 10 M-block column on its own GPU (block-range sharding, no collective
on the data path): weak scaling, value = N * 10.24 G integers / max-over-ranks time.

Prints ONE JSON line on rank 0 (see the task contract): metric/value/unit,
`roofline` (live HIP-event timing of the kernel vs the 8 TB/s HBM peak) and
`cpu_baseline` (the oracle's auto-vectorised C restatement of the reference's
scalar loop, timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_CEILING_GBPS = 6290.0  # same table: measured float4-copy ceiling

WORKLOADS = {
    # name: (type, width, op, bytes per block = SURVEY.md 8(d): 128*W + 128*T [+128 bases])
    "u32_w7_unpack": ("u32", 7, "unpack", 128 * 7 + 128 * 32),
    "u64_w17_unpack": ("u64", 17, "unpack", 128 * 17 + 128 * 64),
    "u64_w17_pack": ("u64", 17, "pack", 128 * 17 + 128 * 64),
    "u32_w12_undelta_pack": ("u32", 12, "undelta_pack", 128 * 12 + 128 + 128 * 32),
    "u32_w7_pack": ("u32", 7, "pack", 128 * 7 + 128 * 32),
    "u16_w3_unpack": ("u16", 3, "unpack", 128 * 3 + 128 * 16),
    # BASELINE.json configs[4]: width[b] = 1 + b % 32; bytes per block averaged over the 32 widths
    "u32_mixed_unpack": ("u32", None, "unpack_mixed", 128 * 16.5 + 128 * 32),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=10_000_000, help="1024-value blocks per GPU")
    ap.add_argument("--workload", default="u32_w7_unpack", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=6.0, help="CPU baseline time budget per leg")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 PMC passes (roofline.traffic "
                    "then comes from profiles/pmc_traffic.json, or is null)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for the barrier / max-time "
                    "reduction (nccl = RCCL; gloo only for plumbing tests)")
    ap.add_argument("--single-device", action="store_true",
                    help="plumbing test: put every rank on cuda:0 (use with --backend gloo)")
    return ap.parse_args()


def rand_u8(nbytes, seed, dev):
    """Uniform random bytes on the device (never zero/constant data: DVFS)."""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    n8 = (nbytes + 7) // 8
    return torch.randint(-2**63, 2**63 - 1, (n8,), dtype=torch.int64, device=dev, generator=g).view(torch.uint8)[:nbytes]


def cpu_baseline(args, ty, width, op):
    """Oracle 'fast' family (C restatement, gcc -O3, lane loop auto-vectorised) on host cores."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from datagen import values
    from oracle_lib import lanes, load_native_oracle, packed_len
    o, cflags = load_native_oracle()
    n = 524288  # blocks: 537 M integers; u32 W=7: 470 MB in + 2.1 GB out (DRAM-resident)
    pl = packed_len(ty, width)
    cores = os.cpu_count() or 1
    npdt = values(ty, 1, 0).dtype
    esz = npdt.itemsize
    in_elems, out_elems = (1024, pl) if op == "pack" else (pl, 1024)
    # random input / output pages first-touched by the thread that will stream them (NUMA placement)
    src = o.parallel_fill(np.empty(n * in_elems, dtype=npdt), in_elems * esz, n, 7, cores)
    out = o.parallel_fill(np.empty(n * out_elems, dtype=npdt), out_elems * esz, n, 9, cores)
    aux = o.parallel_fill(np.empty(n * lanes(ty), dtype=npdt), 128, n, 8, cores) if op == "undelta_pack" else None
    res = {}
    for label, nt in (("single_thread", 1), ("all_cores", cores)):
        o.fast(op, ty, width, src, aux=aux, n_blocks=n, nthreads=nt, out=out)  # warm (page faults)
        best = None
        t_end = time.time() + args.cpu_seconds
        reps = 0
        while time.time() < t_end or reps < 2:
            t0 = time.perf_counter()
            o.fast(op, ty, width, src, aux=aux, n_blocks=n, nthreads=nt, out=out)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            reps += 1
        res[label] = n * 1024 / best / 1e9
    # configs[0] / benches/bitpacking.rs:67-99: the criterion "throughput" shape -- 1024 blocks of
    # u16 W=3, cache-resident, single thread, reported like criterion's Throughput::Bytes(N*2)
    crit = {}
    try:
        nb = 1024
        v16 = (np.arange(nb * 1024) % 8).astype(np.uint16)
        p16 = np.zeros(nb * 192, dtype=np.uint16)
        u16 = np.zeros(nb * 1024, dtype=np.uint16)
        for name, fn in (("compress", lambda: o.fast("pack", "u16", 3, v16, n_blocks=nb, out=p16)),
                         ("decompress", lambda: o.fast("unpack", "u16", 3, p16, n_blocks=nb, out=u16))):
            fn()
            best = None
            for _ in range(200):
                t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            crit[name + "_GBps_of_unpacked_bytes"] = round(nb * 2048 / best / 1e9, 2)
        crit["ok"] = bool(np.array_equal(u16, v16))
    except Exception as e:  # the baseline must never take the bench down
        crit = {"error": str(e)}
    return {
        "criterion_throughput_shape_u16_w3_1024_blocks_1_thread": crit,
        "value": round(res["all_cores"], 3),
        "unit": "Gint/s",
        "cores": cores,
        "kind": "port",
        "single_thread_value": round(res["single_thread"], 3),
        "sample": f"{op} {ty} W={width}, {n} blocks ({n * 1024 / 1e6:.0f} M ints, DRAM-resident), "
                  f"best of repeated passes over ~{args.cpu_seconds:.0f} s per leg; oracle/ C restatement of the "
                  "reference scalar loop (gcc -O3 " + cflags + ", lane loop auto-vectorised), not the Rust crate",
    }


def pmc_child(args):
    """Body of the rocprofv3 --pmc passes: a calibration copy of known size, then 3 launches of the
    workload's kernel.  No timing, no oracle."""
    import torch
    import fastlanes_amd as fl
    dev = torch.device("cuda", 0)
    ty, width, op, _ = WORKLOADS[args.workload]
    tdt = {"u8": torch.uint8, "u16": torch.uint16, "u32": torch.uint32, "u64": torch.uint64}[ty]
    esz = {"u8": 1, "u16": 2, "u32": 4, "u64": 8}[ty]
    a = rand_u8(PMC_CAL_BYTES, 5, dev)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)                                   # known traffic: PMC_CAL_BYTES read + written
    torch.cuda.synchronize()
    del a, b
    n = args.blocks
    if op == "unpack_mixed":
        return
    in_b, out_b = (1024 * esz, 128 * width) if op == "pack" else (128 * width, 1024 * esz)
    src = rand_u8(n * in_b, 6, dev).view(tdt)
    dst = torch.empty(n * out_b // esz, dtype=tdt, device=dev)
    bases = rand_u8(n * 128, 7, dev).view(tdt) if op == "undelta_pack" else None
    for _ in range(3):
        if op == "unpack":
            fl.BitPacking.unpack(width, src, output=dst)
        elif op == "pack":
            fl.BitPacking.pack(width, src, output=dst)
        else:
            fl.Delta.undelta_pack(width, src, bases, output=dst)
    torch.cuda.synchronize()


PMC_CAL_BYTES = 2 << 30


def live_pmc_traffic(args):
    """HBM bytes per launch of the workload's kernel from rocprofv3 PMC counters, collected as
    MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), with
    --kernel-trace only, in KiB units, and FETCH_SIZE corrected by the factor a copy of known size
    shows in the same pass (gfx950 reports half of wide coalesced reads).  None on any failure."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    out = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="fl_pmc_", dir="/tmp")
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", args.workload,
                   "--blocks", str(args.blocks)]
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=120,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            cal, ker = [], []
            for r in csv.DictReader(open(files[0])):
                if r["Counter_Name"] != counter:
                    continue
                v = float(r["Counter_Value"]) * 1024.0
                name = r["Kernel_Name"]
                if "fl::k_" in name:
                    ker.append(v)
                elif v > 0.25 * PMC_CAL_BYTES and ("copy" in name.lower() or "elementwise" in name.lower()) \
                        and "distribution" not in name:
                    cal.append(v)
            shutil.rmtree(d, ignore_errors=True)
            if not ker or not cal:
                return None
            out[counter] = (sum(ker) / len(ker), PMC_CAL_BYTES / (sum(cal) / len(cal)))
        fetch, f_factor = out["FETCH_SIZE"]
        write, w_factor = out["WRITE_SIZE"]
        if not (1.8 < f_factor < 2.2 and 0.9 < w_factor < 1.1):
            return None   # calibration does not look like the documented gfx950 behaviour: do not guess
        return {"bytes": fetch * f_factor + write * w_factor, "fetch_factor": round(f_factor, 4),
                "write_factor": round(w_factor, 4)}
    except Exception:
        return None


def main():
    args = parse()
    if args.pmc_child:
        return pmc_child(args)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.single_device:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    import fastlanes_amd as fl
    fl.load()  # fails loudly if the HIP extension is missing

    ty, width, op, bytes_per_block = WORKLOADS[args.workload]
    n = args.blocks
    tdt = {"u8": torch.uint8, "u16": torch.uint16, "u32": torch.uint32, "u64": torch.uint64}[ty]
    esz = {"u8": 1, "u16": 2, "u32": 4, "u64": 8}[ty]
    un_bytes = 1024 * esz
    plan = None
    if op == "unpack_mixed":
        import numpy as np
        first = rank * n
        if args.blocks == 10_000_000:
            # BASELINE.json configs[4]: 10 B integers = 9 765 625 blocks in total, sharded by block range
            from fastlanes_amd.sharding import block_range
            first, n = block_range(9_765_625, world, rank)
        widths = (1 + (np.arange(n, dtype=np.int64) + first) % 32).astype(np.uint8)
        plan = fl.MixedWidthPlan(ty, widths)
        src = rand_u8(plan.packed_bytes, 1234 + rank, dev).view(tdt)
        dst = torch.empty(n * 1024, dtype=tdt, device=dev)
        in_bytes, out_bytes = plan.packed_bytes / n, un_bytes
        bytes_per_block = in_bytes + out_bytes
    else:
        pl_bytes = 128 * width
        in_bytes, out_bytes = (un_bytes, pl_bytes) if op == "pack" else (pl_bytes, un_bytes)
        src = rand_u8(n * in_bytes, 1234 + rank, dev).view(tdt)
        dst = torch.empty(n * out_bytes // esz, dtype=tdt, device=dev)
    bases = rand_u8(n * 128, 99 + rank, dev).view(tdt) if op == "undelta_pack" else None

    def step():
        if op == "unpack_mixed":
            plan.unpack(src, output=dst)
        elif op == "unpack":
            fl.BitPacking.unpack(width, src, output=dst)
        elif op == "pack":
            fl.BitPacking.pack(width, src, output=dst)
        else:
            fl.Delta.undelta_pack(width, src, bases, output=dst)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    # ---- timed region: barrier + sync on both sides, exactly K steps -----------------
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()   # HIP events on torch's current stream == the stream the kernel is launched on
        step()
        b.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = [a.elapsed_time(b) for a, b in evs]

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- rank 0, N=1: the cpu_baseline leg (the only place bench.py touches oracle/) also
    # ---- checks sampled blocks of what was just timed against the oracle, outside the timed region
    check = None
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and op != "unpack_mixed":
        import numpy as np
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_lib import load_oracle
        o = load_oracle()
        npdt = {"u8": np.uint8, "u16": np.uint16, "u32": np.uint32, "u64": np.uint64}[ty]
        ok = True
        ipb, opb = in_bytes // esz, out_bytes // esz
        for b in sorted({0, 1, 31, 32, n // 2, n - 2, n - 1}):
            s = src[b * ipb:(b + 1) * ipb].view(torch.uint8).cpu().numpy().view(npdt)
            d = dst[b * opb:(b + 1) * opb].view(torch.uint8).cpu().numpy().view(npdt)
            if op == "unpack":
                want = o.unpack(ty, width, s)
            elif op == "pack":
                want = o.pack(ty, width, s)
            else:
                bb = bases[b * (128 // esz):(b + 1) * (128 // esz)].view(torch.uint8).cpu().numpy().view(npdt)
                want = o.undelta_pack(ty, width, s, bb)
            ok = ok and bool(np.array_equal(d, want))
        check = "bit-exact vs oracle on 7 sampled blocks" if ok else "MISMATCH vs oracle"
        cpu = cpu_baseline(args, ty, width, op)

    if rank == 0:
        ints = n * 1024 * world
        if op == "unpack_mixed" and args.blocks == 10_000_000:
            ints = 9_765_625 * 1024    # strong scaling: the whole column, however it is sharded
        value = ints * args.steps / elapsed / 1e9
        avg_kernel_s = sum(kern_ms) / len(kern_ms) / 1e3
        achieved = n * bytes_per_block / avg_kernel_s / 1e9
        traffic = None
        traffic_source = None
        if world == 1 and not args.no_pmc and op != "unpack_mixed":
            live = live_pmc_traffic(args)
            if live is not None:
                traffic = int(live["bytes"])
                traffic_source = ("live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of this "
                                  f"workload; FETCH x{live['fetch_factor']}, WRITE x{live['write_factor']} from a "
                                  f"{PMC_CAL_BYTES >> 30} GiB copy in the same passes")
        prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if traffic is None and os.path.exists(prof):
            traffic_source = "profiles/pmc_traffic.json (committed rocprofv3 PMC run)"
            try:
                traffic = json.load(open(prof)).get(args.workload, {}).get("hbm_bytes_per_launch_at_10M_blocks")
                if traffic is not None and n != 10_000_000:
                    traffic = traffic * n / 10_000_000
            except Exception:
                traffic = None
        out = {
            # BASELINE.json "metric", verbatim, for the headline workload
            "metric": "billion integers/sec decoded (u32 width-7) + achieved HBM GB/s vs peak, 1-8 GPU"
                      if args.workload == "u32_w7_unpack"
                      else f"billion integers/sec ({args.workload})",
            "value": round(value, 2),
            "unit": "Gint/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if op == "unpack_mixed" and args.blocks == 10_000_000 else "weak",
            "vs_baseline": None,
            "dtype": ty,
            "data": "synthetic (uniform random packed bits, generated on device; inputs resident in HBM)",
            "config": {"workload": f"{op} {ty} W={width}, {n} blocks x 1024 values per GPU "
                                   f"(BASELINE.json configs[1])" if args.workload == "u32_w7_unpack"
                                   else (f"{op} {ty} width[b] = 1 + b mod 32 (BASELINE.json configs[4]), {n} blocks per GPU"
                                         if op == "unpack_mixed" else f"{op} {ty} W={width}, {n} blocks per GPU"),
                       "blocks_per_gpu": n, "sharding": "contiguous block range per GPU, no collective"},
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic,
                "traffic_source": traffic_source if traffic is not None else None,
                "frac_of_copy_ceiling": round(achieved / HBM_COPY_CEILING_GBPS, 4),
                "algorithmic_bytes_per_launch": n * bytes_per_block,
                "kernel_ms_avg": round(avg_kernel_s * 1e3, 4),
                "kernel_ms_min": round(min(kern_ms), 4),
                "read_GBps": round(n * in_bytes / avg_kernel_s / 1e9, 1),
                "write_GBps": round(n * out_bytes / avg_kernel_s / 1e9, 1),
                "timing": "HIP events on the launch stream around each of the K launches (rank 0)",
            },
            "correctness": check,
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
        if check and "MISMATCH" in check:
            sys.exit(1)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
