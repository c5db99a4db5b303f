#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X FastLanes decode path.

One "step" = one pass of `BitPacking::unchecked_unpack` (u32, width 7) over a
column of 10 M 1024-value blocks (BASELINE.json configs[1]), input already
resident in HBM, output materialised to HBM.  With --gpus N every rank decodes
its own 10 M-block column on its own GPU (block-range sharding, no collective
on the data path): weak scaling, value = N * 10.24 G integers / max-over-ranks time.

`python bench.py --gpus N` with no torchrun environment SPAWNS the N ranks itself
(one process per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set); under
`python -m torch.distributed.run ... bench.py --gpus N` it is one of the N ranks.
Either way rank 0 prints ONE JSON line.

Control plane (barrier, max-over-ranks time, gather of per-rank figures -- the data
path has NO collective): a gloo group always exists; RCCL ("nccl") is tried on top of
it -- first in a throw-away child process per rank (a hang or crash there costs a
timeout, not the run), then in-process -- and used for the barrier / reductions only if
EVERY rank got it working; otherwise the run continues on gloo.  The line says which
(`control_backend`, `control_fallback_reason`).

Every rank checks what it just timed, both legs, outside the timed region: the first and
last block of ITS slice plus sampled blocks byte for byte against the oracle, then
(--verify full, the default) EVERY block of its slice -- two 64-bit content hashes per
block computed on the device against the multithreaded CPU oracle decoding the same
counter-based stream regenerated on the host (`per_rank[i].correct`,
`per_rank[i].verified_blocks`); any mismatch makes every rank exit non-zero.

Where a workload's buffers live in HBM moves the kernel by a few per cent (DESIGN.md
section 4): the buffers come from the C ABI's own optional helper, fl_column_pair_alloc
(include/fastlanes_amd.h), and --placement auto (default) is its FL_LAYOUT_PROBE: the LIBRARY
allocates every layout it knows -- the one it CONSTRUCTS from measured 1-GiB chunks
(FL_LAYOUT_INTERLEAVED, round 6), plain allocations, the 64-GiB-zoned slab --, times a bare
stream of the pair's read : write proportion on each before anything is filled, keeps the
fastest pair and reports every figure (`roofline.placement_probe_GBps`) -- figures any user of
the header can reproduce.  After
the checks a BARE STREAM of the workload's exact read : write mix is timed on the workload's
own buffers (`roofline.bare_stream_GBps`, `roofline.frac_of_bare_stream`): box and
placement cancel in that ratio, the kernel stays.

After the headline leg the same processes time BASELINE.json configs[4] --
u32, width[b] = 1 + b mod 32, the 10 B-integer column (9 765 625 blocks) sharded
by contiguous block range over the N GPUs, widths / offsets device-resident --
and report it STRONG-scaled under "config5_strong" (per-rank GB/s included).

The line also carries `roofline` (live HIP-event timing of the kernel vs the
8 TB/s HBM peak, PMC traffic) and, at N=1, `cpu_baseline` (the oracle's
auto-vectorised C restatement of the reference's scalar loop, timed on this
box's host cores on a bounded sample).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_CEILING_GBPS = 6290.0  # same table: measured float4-copy ceiling
CONFIG5_BLOCKS = 9_765_625      # 10 B integers / 1024 (BASELINE.json configs[4])

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
    # SURVEY.md 8(d) config 5 "also run a seeded-random width variant": width[b] uniform in 1..32 from a generator seeded with 42,
    # a function of the GLOBAL block index (so every sharding decodes the same column)
    "u32_mixed_random_unpack": ("u32", None, "unpack_mixed", 128 * 16.5 + 128 * 32),
}


def mixed_widths(name, first_block, n, device):
    """widths[first_block .. first_block + n) of a mixed-width workload's column, uint8 on `device`"""
    import torch
    if "random" in name:
        g = torch.Generator(device="cpu")
        g.manual_seed(42)
        total = max(CONFIG5_BLOCKS, first_block + n)
        return torch.randint(1, 33, (total,), dtype=torch.int64, generator=g)[first_block:first_block + n].to(torch.uint8).to(device)
    return (1 + (torch.arange(n, dtype=torch.int64, device=device) + first_block) % 32).to(torch.uint8)
TORCH_DT = {"u8": "uint8", "u16": "uint16", "u32": "uint32", "u64": "uint64"}
ESZ = {"u8": 1, "u16": 2, "u32": 4, "u64": 8}


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
    ap.add_argument("--no-config5", action="store_true", help="skip the strong-scaled mixed-width leg")
    ap.add_argument("--no-dispatch-check", action="store_true", help="skip fl_internal_selftune_check after the timed region (it launches the "
                    "workload's own kernel template on a slice of the buffers: under rocprofv3 --stats those launches would dilute the kernel's average)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--backend", default="auto", choices=("auto", "nccl", "gloo"),
                    help="control plane for the barrier / max-time reduction: auto = RCCL if every rank gets it working, "
                         "else gloo; nccl = the same (the fallback still applies, the line reports it); gloo = never try RCCL")
    ap.add_argument("--no-check", action="store_true", help="skip the per-rank oracle check of the timed output")
    ap.add_argument("--placement", default="auto", choices=("auto", "interleaved", "zoned", "separate"),
                    help="where a workload's buffers live in HBM moves every streaming kernel by a few per cent (DESIGN.md section 4). "
                         "The buffers come from fl_column_pair_alloc (include/fastlanes_amd.h).  auto (default) = FL_LAYOUT_PROBE: the library "
                         "allocates every layout below, times a bare stream on each before anything is filled and keeps the fastest pair, "
                         "every figure is reported (roofline.placement_probe_GBps); "
                         "interleaved: constructed from 1-GiB physical chunks whose class of memory the library measures (input in one "
                         "class, the output's chunks arranged for the eight XCDs' write positions); "
                         "separate: one hipMalloc per buffer, wherever the driver puts it; "
                         "zoned: input and output carved from one allocation, the output centred on a 64-GiB multiple")
    ap.add_argument("--verify", default="auto", choices=("auto", "full", "sample"),
                    help="what every rank checks of what it just timed, outside the timed region: sample = the first / last / sampled "
                         "blocks of its slice against the oracle; full = that, plus two 64-bit content hashes per block of its WHOLE "
                         "slice against the multithreaded CPU oracle decoding the same counter-based stream regenerated on the host "
                         "(decode workloads) or a device-side round trip of the whole slice (pack); auto = full")
    ap.add_argument("--probe-nccl", action="store_true", help="with --dry-run: still attempt the RCCL probe (exercises the "
                    "fallback on a box without GPUs)")
    ap.add_argument("--inject-mismatch", type=int, default=-1, help="with --dry-run: pretend this rank's oracle check failed "
                    "(tests that one bad rank fails the whole launch)")
    ap.add_argument("--nccl-probe-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--nccl-probe-timeout", type=float, default=120.0, help="seconds the throw-away RCCL probe may take")
    ap.add_argument("--single-device", action="store_true",
                    help="plumbing test: put every rank on cuda:0 (use with --backend gloo)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher plumbing only, no GPU: spawn / rendezvous / barrier / max-reduce / sharding run for "
                         "real, the codec step is skipped and the line says so (value null) -- never a measurement")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------
# launcher: `bench.py --gpus N` without a torchrun environment spawns its own N ranks
# ---------------------------------------------------------------------------------------------
def spawn_ranks(args):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0 and rc == 0:
                rc = r
                for q in live:            # a dead rank leaves the others in a barrier: stop exactly our children
                    q.terminate()
    return rc


def ctypes_stream(dev):
    import ctypes
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def rand_u8(nbytes, seed, dev):
    """Uniform random bytes on the device (never zero/constant data: DVFS)."""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    n8 = (nbytes + 7) // 8
    return torch.randint(-2**63, 2**63 - 1, (n8,), dtype=torch.int64, device=dev, generator=g).view(torch.uint8)[:nbytes]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(args, ty, width, op):
    """Oracle 'fast' family (C restatement, gcc -O3, lane loop auto-vectorised) on host cores."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from datagen import values
    from oracle_lib import lanes, load_native_oracle, packed_len
    o, cflags = load_native_oracle()
    n = 524288  # blocks: 537 M integers; u32 W=7: 470 MB in + 2.1 GB out (DRAM-resident)
    cores = os.cpu_count() or 1
    res = {}
    if op == "unpack_mixed":
        # the reference's caller loop over per-block widths (bitpacking.rs:109-129), width[b] = 1 + b mod 32
        widths = mixed_widths(args.workload, 0, n, "cpu").numpy()
        off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(widths.astype(np.uint64) * np.uint64(128), out=off[1:])
        # one fill per 32-block period keeps every thread's pages first-touched by (roughly) its own range
        src = o.parallel_fill(np.empty(int(off[-1]) // 4, dtype=np.uint32), 8, int(off[-1]) // 8, 7, cores)
        out = o.parallel_fill(np.empty(n * 1024, dtype=np.uint32), 4096, n, 9, cores)

        def run(nt):
            o.fast_unpack_mixed_u32(widths, off[:-1], src, n_blocks=n, nthreads=nt, out=out)
        what = (f"unpack u32 width[b] = {'seeded-random in 1..32' if 'random' in args.workload else '1 + b mod 32'} "
                f"(caller loop over per-block widths), {n} blocks")
    else:
        pl = packed_len(ty, width)
        npdt = values(ty, 1, 0).dtype
        esz = npdt.itemsize
        in_elems, out_elems = (1024, pl) if op == "pack" else (pl, 1024)
        # random input / output pages first-touched by the thread that will stream them (NUMA placement)
        src = o.parallel_fill(np.empty(n * in_elems, dtype=npdt), in_elems * esz, n, 7, cores)
        out = o.parallel_fill(np.empty(n * out_elems, dtype=npdt), out_elems * esz, n, 9, cores)
        aux = o.parallel_fill(np.empty(n * lanes(ty), dtype=npdt), 128, n, 8, cores) if op == "undelta_pack" else None

        def run(nt):
            o.fast(op, ty, width, src, aux=aux, n_blocks=n, nthreads=nt, out=out)
        what = f"{op} {ty} W={width}, {n} blocks"
    for label, nt in (("single_thread", 1), ("all_cores", cores)):
        run(nt)  # warm (page faults)
        best = None
        t_end = time.time() + args.cpu_seconds
        reps = 0
        while time.time() < t_end or reps < 2:
            t0 = time.perf_counter()
            run(nt)
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
        "cpu_model": cpu_model(),
        "kind": "port",
        "single_thread_value": round(res["single_thread"], 3),
        "sample": f"{what} ({n * 1024 / 1e6:.0f} M ints, DRAM-resident), "
                  f"best of repeated passes over ~{args.cpu_seconds:.0f} s per leg; oracle/ C restatement of the "
                  "reference scalar loop (gcc -O3 " + cflags + ", lane loop auto-vectorised), not the Rust crate",
    }


# ---------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------
class Workload:
    """One rank's share of a workload: device buffers + step()."""

    def __init__(self, name, n, first_block, rank, dev, placement="separate"):
        """placement: the layout asked of fl_column_pair_alloc (include/fastlanes_amd.h) -- "separate" = one hipMalloc per buffer,
        wherever the driver puts them; "zoned" = input and output carved from ONE allocation, the input at offset 0, the output
        centred on a 64-GiB multiple; "interleaved" = constructed from measured 1-GiB chunks; "auto" = the library tries all of them,
        times a bare stream of in_bytes : out_bytes on each and keeps the fastest pair (self.probe = every figure) -- or "torch" = plain
        torch tensors (several ranks on one device)."""
        import torch
        import fastlanes_amd as fl
        from fastlanes_amd import placement as pl
        self.name = name
        self.ty, self.width, self.op, _ = WORKLOADS[name]
        self.n = n
        ty, width, op = self.ty, self.width, self.op
        tdt = getattr(torch, TORCH_DT[ty])
        esz = ESZ[ty]
        un_bytes = 1024 * esz
        self.bases = None
        self.pair = None
        self.probe = None
        lib = fl.load()

        def buffers(in_bytes, out_bytes, aux_bytes=0):
            """(input, aux, output) as uint8 tensors; the input (and aux) filled with counter-based random bytes on the device"""
            if placement == "torch":
                src = torch.empty(in_bytes, dtype=torch.uint8, device=dev)
                aux = torch.empty(aux_bytes, dtype=torch.uint8, device=dev)
                dst = torch.empty(out_bytes, dtype=torch.uint8, device=dev)
                self.placement = "torch"
            else:
                # through the C ABI's own helper (include/fastlanes_amd.h: fl_column_pair_alloc): "auto" = every layout probed with a
                # bare stream INSIDE the library, the fastest pair kept -- before the buffers are filled
                self.pair = pl.ColumnPair(in_bytes, out_bytes, dev, aux_bytes, layout=placement)
                src, aux, dst = self.pair.input, self.pair.aux, self.pair.output
                self.placement, self.probe = self.pair.layout, self.pair.probe_GBps
                self.classes = self.pair.classes
            st = ctypes_stream(dev)
            self.src_seed, self.aux_seed = 1234 + rank, 99 + rank
            for t, seed in ((src, self.src_seed), (aux, self.aux_seed)):
                nb = t.numel() & ~7
                if nb and lib.fl_fill_random(t.data_ptr(), nb, seed, st) != 0:
                    raise RuntimeError("fl_fill_random failed")
            return src, aux, dst

        if op == "unpack_mixed":
            # widths and offsets are DEVICE arrays, built on the device: nothing about the column touches the host
            self.widths = mixed_widths(name, first_block, n, dev)
            self.offsets, total = fl.widths_to_offsets(ty, self.widths)
            packed_bytes = int(total.item())
            src, _, dst = buffers(packed_bytes, n * un_bytes)
            self.src, self.dst = src.view(tdt), dst.view(tdt)
            self.in_bytes, self.out_bytes = packed_bytes, n * un_bytes          # per launch
            self.step = lambda: fl.unpack_widths(self.widths, self.offsets, self.src, output=self.dst, check=False)
        else:
            pl_bytes = 128 * width
            ib, ob = (un_bytes, pl_bytes) if op == "pack" else (pl_bytes, un_bytes)
            src, aux, dst = buffers(n * ib, n * ob, n * 128 if op == "undelta_pack" else 0)
            self.src, self.dst = src.view(tdt), dst.view(tdt)
            self.in_bytes, self.out_bytes = n * ib, n * ob
            if op == "undelta_pack":
                self.bases = aux.view(tdt)
                self.in_bytes += n * 128
                self.step = lambda: fl.Delta.undelta_pack(width, self.src, self.bases, output=self.dst)
            elif op == "unpack":
                self.step = lambda: fl.BitPacking.unpack(width, self.src, output=self.dst)
            else:
                self.step = lambda: fl.BitPacking.pack(width, self.src, output=self.dst)
        self.bytes = self.in_bytes + self.out_bytes      # algorithmic bytes per launch

    def check_against_oracle(self):
        """The first and last block of this rank's slice plus sampled blocks of what was just timed vs the oracle
        (every rank, outside the timed region; SURVEY.md 8(d) "first/last block of every GPU slice").
        Returns (all equal, number of blocks compared)."""
        import numpy as np
        import torch
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_lib import load_oracle
        o = load_oracle()
        ty, width, op, n = self.ty, self.width, self.op, self.n
        esz = ESZ[ty]
        npdt = {"u8": np.uint8, "u16": np.uint16, "u32": np.uint32, "u64": np.uint64}[ty]

        def host(t):
            return t.view(torch.uint8).cpu().numpy().view(npdt)
        ok = True
        blocks = sorted(b for b in {0, 1, 31, 32, n // 3, n // 2, n - 2, n - 1} if 0 <= b < n)
        for b in blocks:
            if op == "unpack_mixed":
                w = int(self.widths[b].item())
                lo = int(self.offsets[b].item()) // esz
                want = o.unpack(ty, w, host(self.src[lo:lo + 1024 * w // (8 * esz)]))
                got = host(self.dst[b * 1024:(b + 1) * 1024])
            else:
                ipb, opb = self.src.numel() // n, self.dst.numel() // n
                s = host(self.src[b * ipb:(b + 1) * ipb])
                got = host(self.dst[b * opb:(b + 1) * opb])
                if op == "unpack":
                    want = o.unpack(ty, width, s)
                elif op == "pack":
                    want = o.pack(ty, width, s)
                else:
                    want = o.undelta_pack(ty, width, s, host(self.bases[b * (128 // esz):(b + 1) * (128 // esz)]))
            ok = ok and bool(np.array_equal(got, want))
        return ok, len(blocks)


    def verify_full(self, threads, chunk=250_000):
        """EVERY block of this rank's slice (SURVEY.md 8(d) "correctness at scale" (2)), outside the timed region.
        Decode workloads: the input is a counter-based stream (fl_fill_random), so the host regenerates it chunk by chunk without a
        PCIe transfer, the multithreaded CPU oracle decodes it, and two 64-bit content hashes per block (sum and position-weighted
        sum, wrapping) of what the GPU wrote must equal the oracle's.  pack: unpack(pack(x)) == x & mask for the whole slice on
        the device (the packed bytes themselves are compared with the oracle on the sampled blocks).
        Returns (all equal, number of blocks verified)."""
        import numpy as np
        import torch
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_lib import load_oracle
        import fastlanes_amd as fl
        o = load_oracle()
        ty, width, op, n = self.ty, self.width, self.op, self.n
        esz = ESZ[ty]
        npdt = {"u8": np.uint8, "u16": np.uint16, "u32": np.uint32, "u64": np.uint64}[ty]
        GOLDEN, M64 = 0x9E3779B97F4A7C15, (1 << 64) - 1
        dev = self.dst.device
        w_idx = torch.arange(1, 1025, dtype=torch.int64, device=dev)

        def regenerated(seed, first_word, n_words):
            """64-bit words [first_word, first_word + n_words) of fl_fill_random's stream `seed`, on the host"""
            a = np.empty(n_words, dtype=np.uint64)
            o.parallel_fill(a, 8, n_words, (seed * GOLDEN + first_word * GOLDEN) & M64, threads)
            return a

        def device_hashes(t):
            """(sum, position-weighted sum) per 1024-value block of a device tensor, wrapping uint64"""
            k = t.numel() // 1024
            if ty == "u64":
                vals = t.view(torch.int64).view(k, 1024)
            elif ty == "u32":
                vals = t.view(torch.int32).view(k, 1024).to(torch.int64) & 0xFFFFFFFF
            elif ty == "u16":
                vals = t.view(torch.int16).view(k, 1024).to(torch.int64) & 0xFFFF
            else:
                vals = t.view(torch.uint8).view(k, 1024).to(torch.int64)
            return vals.sum(dim=1).cpu().numpy().view(np.uint64), (vals * w_idx).sum(dim=1).cpu().numpy().view(np.uint64)

        verified = 0
        if op == "pack":
            mask = (1 << width) - 1
            for k0 in range(0, n, chunk):
                nb = min(chunk, n - k0)
                pl = 1024 * width // (8 * esz)
                back = fl.BitPacking.unpack(width, self.dst[k0 * pl:(k0 + nb) * pl])
                want = self.src[k0 * 1024:(k0 + nb) * 1024]
                if width < 8 * esz:
                    want = (want.view(getattr(torch, TORCH_DT[ty].replace("u", ""))) & (mask if mask < (1 << 63) else -1)).view(want.dtype)
                if not torch.equal(back.view(torch.uint8), want.view(torch.uint8)):
                    return False, verified
                verified += nb
                del back, want
            return True, verified
        if op == "unpack_mixed":
            offsets = self.offsets.cpu().numpy().view(np.uint64)
            widths = self.widths.cpu().numpy()
            total = int(self.in_bytes)
        for k0 in range(0, n, chunk):
            nb = min(chunk, n - k0)
            if op == "unpack_mixed":
                b0 = int(offsets[k0])
                b1 = int(offsets[k0 + nb]) if k0 + nb < n else total
                host_pk = regenerated(self.src_seed, b0 // 8, (b1 - b0) // 8).view(np.uint32)
                rel = (offsets[k0:k0 + nb] - np.uint64(b0)).astype(np.uint64)
                host_out = o.fast_unpack_mixed_u32(widths[k0:k0 + nb], rel, host_pk, nthreads=threads)
            else:
                wpb = 16 * width                                    # 64-bit words per packed block
                host_pk = regenerated(self.src_seed, k0 * wpb, nb * wpb).view(npdt)
                if op == "undelta_pack":
                    host_bases = regenerated(self.aux_seed, k0 * 16, nb * 16).view(npdt)
                    host_out = o.fast(op, ty, width, host_pk, aux=host_bases, n_blocks=nb, nthreads=threads)
                else:
                    host_out = o.fast("unpack", ty, width, host_pk, n_blocks=nb, nthreads=threads)
            s_cpu, w_cpu = o.block_hashes(ty, host_out, threads)
            s_gpu, w_gpu = device_hashes(self.dst[k0 * 1024:(k0 + nb) * 1024])
            if not (np.array_equal(s_gpu, s_cpu) and np.array_equal(w_gpu, w_cpu)):
                return False, verified
            verified += nb
            del host_pk, host_out
        return True, verified


def check_text(flags, n_checked, verified=None):
    if not flags:
        return None
    if all(flags):
        t = f"bit-exact vs oracle on {n_checked} blocks per rank (first, last and sampled blocks of every rank's slice)"
        if verified:
            t += (f"; every block of every rank's slice verified ({sum(verified)} blocks in all: content hashes vs the CPU oracle "
                  "decoding the regenerated stream, or a device round trip for pack)")
        return t
    return "MISMATCH vs oracle on rank(s) " + ",".join(str(r) for r, f in enumerate(flags) if not f)


# ---------------------------------------------------------------------------------------------
# control plane: gloo always, RCCL on top of it when every rank gets it working
# ---------------------------------------------------------------------------------------------
def nccl_probe_child(args):
    """Throw-away process: RCCL rendezvous + one all_reduce + one barrier on this rank's device.  Exit 0 = worked.
    Run by Control.__init__ under a timeout, so an RCCL hang or crash never reaches the measuring process."""
    import datetime
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    local = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=args.nccl_probe_timeout))
    t = torch.ones(1, device=dev)
    dist.all_reduce(t)
    dist.barrier(device_ids=[local])
    torch.cuda.synchronize()
    ok = int(t.item()) == world
    os._exit(0 if ok else 3)          # no destructors: a half-dead communicator must not hang the exit


class stdout_to_stderr:
    """Redirect file descriptor 1 to 2 for the duration (C++ libraries that print to stdout)."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


class Control:
    """barrier / max / gather over the ranks.  The data path needs none of it, so RCCL is an
    option, never a requirement: `backend` is what is actually in use."""

    def __init__(self, args, world, rank, dev):
        import datetime
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world, self.rank, self.dev = world, rank, dev
        self.backend, self.group, self.fallback_reason, self.inprocess_failed = "none", None, None, False
        if world == 1:
            return
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with stdout_to_stderr():          # gloo announces its connections on stdout; stdout carries exactly ONE JSON line
            dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=30))
            dist.barrier()
        self.backend = "gloo"
        if args.backend == "gloo" or (args.dry_run and not args.probe_nccl):
            return
        # 1. probe RCCL in a child process per rank (own rendezvous port), bounded by a timeout
        err = None
        port = torch.zeros(1, dtype=torch.int64)
        if rank == 0:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port[0] = sk.getsockname()[1]
        dist.broadcast(port, 0)
        try:
            # its own rendezvous: a fresh port, and none of torchrun's TORCHELASTIC_* variables (with
            # TORCHELASTIC_USE_AGENT_STORE every rank would look for the agent's store on that port instead of rank 0 hosting one)
            env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
            env["MASTER_PORT"] = str(int(port.item()))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            cmd = [sys.executable, os.path.abspath(__file__), "--nccl-probe-child", "--gpus", str(world),
                   "--nccl-probe-timeout", str(args.nccl_probe_timeout)] + (["--single-device"] if args.single_device else [])
            r = subprocess.run(cmd, env=env, timeout=args.nccl_probe_timeout + 60, capture_output=True, text=True)
            if r.returncode != 0:
                lines = [l.strip() for l in (r.stderr or "").splitlines() if l.strip() and "destroy_process_group" not in l]
                # the exception line, plus what RCCL printed after "Last error:" (e.g. "Duplicate GPU detected ...")
                exc = [l for l in lines if "Error" in l and not l.startswith(("File ", "Traceback"))]
                last = [lines[i + 1] for i, l in enumerate(lines[:-1]) if l.startswith("Last error")]
                telling = " / ".join((exc[-1:] + last[-1:])) or (lines[-1] if lines else "no stderr")
                err = f"RCCL probe exited {r.returncode}: {telling[:300]}"
        except subprocess.TimeoutExpired:
            err = f"RCCL probe timed out after {args.nccl_probe_timeout + 60:.0f} s"
        except Exception as e:       # the probe must never take the bench down
            err = f"RCCL probe could not run: {e!r}"
        if not self._all_ok(err is None):
            self.fallback_reason = err or "the RCCL probe failed on another rank"
            return
        # 2. the probe worked everywhere: the same thing in-process, still guarded
        try:
            os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")     # a timeout raises instead of aborting the process
            with stdout_to_stderr():
                g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=args.nccl_probe_timeout))
                t = torch.ones(1, device=dev)
                dist.all_reduce(t, group=g)
                torch.cuda.synchronize()
            if int(t.item()) != world:
                raise RuntimeError(f"all_reduce over RCCL returned {t.item()} for {world} ranks")
        except Exception as e:
            g, err, self.inprocess_failed = None, f"in-process RCCL init failed: {e!r}"[:300], True
        if self._all_ok(g is not None):
            self.group, self.backend = g, "nccl"
        else:
            self.fallback_reason = err or "in-process RCCL init failed on another rank"

    def _all_ok(self, ok):
        """logical AND over the ranks, on the gloo plane"""
        t = self.torch.tensor([1.0 if ok else 0.0], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return t.item() == 1.0

    def barrier(self):
        if self.world == 1:
            return
        if self.group is not None:
            self.dist.barrier(group=self.group, device_ids=[self.dev.index])
        else:
            self.dist.barrier()

    def _tensor(self, vals):
        on_gpu = self.group is not None
        return self.torch.tensor(vals, dtype=self.torch.float64, device=self.dev if on_gpu else "cpu")

    def all_max(self, x):
        if self.world == 1:
            return x
        t = self._tensor([x])
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def gather(self, vals):
        """[[vals of rank 0], [vals of rank 1], ...] on every rank."""
        t = self._tensor(vals)
        if self.world == 1:
            return [t.tolist()]
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.group)
        return [o.tolist() for o in out]

    def close(self):
        if self.world == 1:
            return
        self.dist.barrier()                       # gloo: everybody is done
        if self.inprocess_failed:
            return                                # a failed RCCL attempt may have left threads behind: main() leaves by os._exit
        try:
            self.dist.destroy_process_group()
        except Exception:
            pass


def timed_region(step, args, ctl):
    """W untimed warm-up steps, then EXACTLY K steps bracketed by barrier + synchronize on both sides;
    returns (max-over-ranks wall seconds, this rank's per-launch HIP-event milliseconds)."""
    import torch
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    ctl.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()   # HIP events on torch's current stream == the stream the kernel is launched on
        step()
        b.record()
    torch.cuda.synchronize()
    ctl.barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = [a.elapsed_time(b) for a, b in evs]
    return ctl.all_max(elapsed), kern_ms


# ---------------------------------------------------------------------------------------------
# PMC traffic (N=1): the workload's kernel re-run under rocprofv3, separate passes per counter
# ---------------------------------------------------------------------------------------------
PMC_CAL_BYTES = 2 << 30


def pmc_child(args):
    """Body of the rocprofv3 --pmc passes: a calibration copy of known size, then 3 launches of the
    workload's kernel.  No timing, no oracle."""
    import torch
    dev = torch.device("cuda", 0)
    a = rand_u8(PMC_CAL_BYTES, 5, dev)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)                                   # known traffic: PMC_CAL_BYTES read + written
    torch.cuda.synchronize()
    del a, b
    n = CONFIG5_BLOCKS if (WORKLOADS[args.workload][2] == "unpack_mixed" and args.blocks == 10_000_000) else args.blocks
    w = Workload(args.workload, n, 0, 0, dev, "separate")     # HBM traffic does not depend on where the buffers live
    for _ in range(3):
        w.step()
    torch.cuda.synchronize()


def live_pmc_traffic(args, workload=None):
    """HBM bytes per launch of the workload's kernel from rocprofv3 PMC counters, collected as
    MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), with
    --kernel-trace only, in KiB units, and FETCH_SIZE corrected by the factor a copy of known size
    shows in the same pass (gfx950 reports half of wide coalesced reads).  None on any failure."""
    import csv
    import glob
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    out = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="fl_pmc_", dir="/tmp")
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", workload or args.workload,
                   "--blocks", str(args.blocks if workload is None else 10_000_000)]
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
            subprocess.run(cmd, cwd="/tmp", env=dict(env, TMPDIR="/tmp"), timeout=180,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            cal, ker = [], []
            for r in csv.DictReader(open(files[0])):
                if r["Counter_Name"] != counter:
                    continue
                v = float(r["Counter_Value"]) * 1024.0
                name = r["Kernel_Name"]
                if "fl::k_" in name and "k_scan" not in name and "k_fill" not in name:
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


def roofline(w, kern_ms, traffic=None, traffic_source=None, bare=None):
    avg_s = sum(kern_ms) / len(kern_ms) / 1e3
    med_ms = sorted(kern_ms)[len(kern_ms) // 2]
    achieved = w.bytes / avg_s / 1e9
    extra = {}
    if bare:
        # the kernel against a bare stream of its own bytes on its own buffers in this run: box and placement cancel, the kernel stays
        extra = {"bare_stream_GBps": bare["GBps"], "bare_stream_best_GBps": bare["best_GBps"],
                 "frac_of_bare_stream": round(w.bytes / med_ms / 1e6 / bare["GBps"], 4),
                 "bare_stream_frac_of_peak": round(bare["GBps"] / HBM_PEAK_GBPS, 4),
                 "bare_stream": bare["shape"] + "; median / best of 9 launches after the timed region, same buffers; "
                                "frac_of_bare_stream = median kernel rate / median stream rate"}
    return dict({
        "bound": "hbm",
        "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 4),
        "traffic": traffic,
        "traffic_source": traffic_source if traffic is not None else None,
        "frac_of_copy_ceiling": round(achieved / HBM_COPY_CEILING_GBPS, 4),
        "algorithmic_bytes_per_launch": int(w.bytes),
        "kernel_ms_avg": round(avg_s * 1e3, 4),
        "kernel_ms_median": round(med_ms, 4),
        "kernel_ms_min": round(min(kern_ms), 4),
        "read_GBps": round(w.in_bytes / avg_s / 1e9, 1),
        "write_GBps": round(w.out_bytes / avg_s / 1e9, 1),
        "timing": "HIP events on the launch stream around each of the K launches (rank 0); achieved / frac from the average, "
                  "median and min beside it",
    }, **extra)


def run_check(w, args, ctl):
    """Per-rank check of what was just timed; (flags of every rank, blocks compared with the oracle per rank, blocks fully verified
    per rank) or ([], 0, []) with --no-check."""
    if args.no_check:
        return [], 0, []
    verified = 0
    try:
        ok, n_checked = w.check_against_oracle()
        if ok and args.verify != "sample":
            local = int(os.environ.get("LOCAL_WORLD_SIZE", str(ctl.world)))
            ok, verified = w.verify_full(max(1, min(64, (os.cpu_count() or 1) // max(1, local))))
    except Exception as e:          # an unloadable checker is a failed check, not a crash that strands the other ranks
        print(f"rank {ctl.rank}: oracle check could not run: {e!r}", file=sys.stderr)
        ok, n_checked = False, 0
    got = ctl.gather([1.0 if ok else 0.0, float(verified)])
    return [bool(v[0]) for v in got], n_checked, [int(v[1]) for v in got]


PLACEMENT_TEXT = {
    "zoned": "fl_column_pair_alloc(FL_LAYOUT_ZONED): input and output carved from ONE allocation, the input at offset 0, the output centred "
             "on the 64-GiB multiple behind it (include/fastlanes_amd.h, DESIGN.md section 4)",
    "separate": "fl_column_pair_alloc(FL_LAYOUT_SEPARATE): one hipMalloc per buffer, wherever the driver puts it",
    "interleaved": "fl_column_pair_alloc(FL_LAYOUT_INTERLEAVED): CONSTRUCTED from 1-GiB physical chunks (hipMemCreate / hipMemMap) whose class "
                   "of memory the library measured -- the input inside one class, the output's chunks arranged so that the eight XCDs' write positions spread over the classes at every moment "
                   "(include/fastlanes_amd.h, DESIGN.md section 4)",
    "torch": "one torch allocation per buffer (several ranks share one device)",
}


def placement_text(placed, probe, classes=""):
    t = PLACEMENT_TEXT[placed]
    if placed == "interleaved" and classes:
        t += f"; measured class of every chunk, input first: {classes}"
    if probe:
        t += ("; chosen by fl_column_pair_alloc(FL_LAYOUT_PROBE) -- the library's own measurement, reproducible through the header alone: a "
              "bare stream of the pair's read : write proportion on each candidate layout before the buffers were filled, GB/s " +
              ", ".join(f"{k} {v}" for k, v in probe.items()))
    return t


def release(w):
    """free a workload's device buffers NOW (w.step closes over w: without breaking the cycle they would outlive `del w`)"""
    import gc
    import torch
    if w is not None:
        w.step = w.src = w.dst = w.bases = None
        if w.pair is not None:
            w.pair.free()
            w.pair = None
    gc.collect()
    torch.cuda.empty_cache()


def bare_stream(w, launches=9):
    """A BARE STREAM of the workload's read : write mix on the workload's OWN buffers, in the same run (fl_internal_bare_stream with the
    launch shape the library's kernel for this call has: fastlanes_amd_internal.h): what the memory gives these bytes at these addresses
    on this box, with no codec work at all.  Overwrites the output: only after the checks.  {"GBps": median, "best_GBps": .., ...}"""
    import ctypes
    import torch
    import fastlanes_amd as fl
    lib = fl.load()
    op = {"unpack": 0, "pack": 1, "undelta_pack": 2, "unpack_mixed": 3}[w.op]
    Z, I = ctypes.c_size_t, ctypes.c_int
    iu, au, ou, nt, wv, wn, bpu = Z(), Z(), Z(), I(), I(), I(), ctypes.c_uint()
    if lib.fl_internal_bare_stream_shape(op, 8 * ESZ[w.ty], 33 if op == 3 else w.width, *[ctypes.byref(x) for x in (iu, au, ou, nt, wv, wn, bpu)]) != 0:
        return None
    if getattr(w, "pair", None) is not None and getattr(w.pair, "classes", ""):
        wn.value = 31                # buffers inside a constructed pair: the library launches under the whole-column tile map (fl_kernels.hpp)
    n = w.n // bpu.value             # units of bpu consecutive blocks (u8: 4, u16: 2): a wavefront's share in the library's kernels too
    in_unit = iu.value
    if op == 3:                      # a mixed-width column: units of the column's mean packed block, rounded down to a cell
        in_unit = (w.in_bytes // n) & ~15
    aux = w.bases.data_ptr() if au.value else None
    st = ctypes_stream(w.dst.device)

    def launch():
        return lib.fl_internal_bare_stream(w.src.data_ptr(), in_unit, aux, au.value, w.dst.data_ptr(), ou.value, n, nt.value, wv.value, wn.value, st)
    for _ in range(2):
        if launch() != 0:
            return None
    torch.cuda.synchronize()
    ms = []
    for _ in range(launches):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); launch(); b.record(); b.synchronize()
        ms.append(a.elapsed_time(b))
    ms.sort()
    nbytes = n * (in_unit + au.value + ou.value)
    return {"GBps": round(nbytes / ms[len(ms) // 2] / 1e6, 1), "best_GBps": round(nbytes / ms[0] / 1e6, 1), "bytes": nbytes,
            "shape": f"{in_unit} B read{' + %d B aux' % au.value if au.value else ''} : {ou.value} B written per wavefront, "
                     f"{'non-temporal' if nt.value else 'default'} loads, {wv.value} waves/SIMD, "
                     f"{'whole-column tile map' if wn.value >= 31 else '2^%d-block windows' % wn.value} (the shape of the library's kernel for this call)"}


def dispatch_check(w, gib=1.0):
    """Is the generated dispatch table's choice for this call a loser on THIS box?  fl_internal_selftune_check (fastlanes_amd_internal.h)
    times the table's kernel against every alternative the library could launch -- the cell-column kernel where it is built, the
    wave-per-block kernel at 3 / 4 / 5 / 6 / 8 wavefronts per SIMD -- on about `gib` GiB of the workload's own buffers, after the timed region
    and the checks (it overwrites the output).  Reported, never acted on: what gets timed is the shipped table."""
    import ctypes
    import fastlanes_amd as fl
    lib = fl.load()
    op = {"unpack": 0, "pack": 1, "undelta_pack": 2}.get(w.op)
    if op is None:
        return None
    n = max(1024, min(w.n, int(gib * (1 << 30) / WORKLOADS[w.name][3])))
    t, o, pol = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_int(0)
    aux = w.bases.data_ptr() if w.bases is not None else None
    rc = lib.fl_internal_selftune_check(op, 8 * ESZ[w.ty], w.width, w.src.data_ptr(), aux, w.dst.data_ptr(), n, ctypes_stream(w.dst.device),
                                        ctypes.byref(t), ctypes.byref(o), ctypes.byref(pol))
    if rc != 0 or t.value <= 0:
        return None
    other = "none" if not pol.value else "cell-column" if pol.value == 1 else f"wave-per-block at {pol.value >> 8} waves/SIMD"
    behind = (t.value / o.value - 1) * 100 if o.value > 0 else 0.0
    return {"blocks": n, "table_ms": round(t.value, 4), "best_alternative_ms": round(o.value, 4), "best_alternative": other,
            "table_behind_pct": round(behind, 2), "table_loses_by_more_than_3pct": bool(behind > 3.0),
            "note": "fl_internal_selftune_check on ~%.0f GiB of this workload's buffers after the timed region; the table "
                    "(fl_dispatch_table.inc) was measured at BASELINE sizes -- small launches favour other occupancies" % gib}


def placed_workload(name, n, first, rank, dev, args, single_device=False):
    """(workload, the placement probe's figures or None).  Where a column lives in HBM moves the same kernel by a few per cent,
    differently per workload and per box, and nothing in an address tells: --placement auto asks the LIBRARY
    (fl_column_pair_alloc, FL_LAYOUT_PROBE) to allocate both layouts it knows, time a bare stream on each and keep the faster pair --
    a measurement made before the buffers are filled, by a call any user of the header can make; both figures go into the line."""
    if single_device:               # several ranks on one device: no room for a probe each
        return Workload(name, n, first, rank, dev, "torch"), None
    w = Workload(name, n, first, rank, dev, args.placement)
    return w, w.probe


def config5_leg(args, world, rank, dev, ctl):
    """BASELINE.json configs[4], STRONG scaling: the 10 B-integer u32 column (9 765 625 blocks, width[b] = 1 + b mod 32)
    sharded by contiguous block range over the ranks (no collective on the data path).  `w` = this rank's slice, already
    resident in HBM (main() builds both legs' columns before anything is timed)."""
    from fastlanes_amd.sharding import block_range
    first, n = block_range(CONFIG5_BLOCKS, world, rank)
    if args.dry_run:
        per_rank = ctl.gather([float(n), 0.0, 0.0])
        return {"dry_run": True, "per_rank": [{"rank": r, "first_block": block_range(CONFIG5_BLOCKS, world, r)[0],
                                               "blocks": int(v[0])} for r, v in enumerate(per_rank)]}
    w, probe = placed_workload("u32_mixed_unpack", n, first, rank, dev, args, args.single_device)
    elapsed, kern_ms = timed_region(w.step, args, ctl)
    avg_ms = sum(kern_ms) / len(kern_ms)
    per_rank = ctl.gather([float(n), avg_ms, float(w.bytes)])
    flags, n_checked, verified = run_check(w, args, ctl)
    if rank != 0:
        release(w)
        return {"flags": flags}
    traffic = source = None
    placed, classes = w.placement, getattr(w, "classes", "")
    bare = bare_stream(w)                    # after the checks (it overwrites the output), same buffers, same run
    release(w)                               # measured and checked (the PMC child builds its own copy of the column)
    if world == 1 and not args.no_pmc:
        live = live_pmc_traffic(args, "u32_mixed_unpack")
        if live is not None:
            traffic = int(live["bytes"])
            source = ("live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of this workload; "
                      f"FETCH x{live['fetch_factor']}, WRITE x{live['write_factor']} from a {PMC_CAL_BYTES >> 30} GiB copy in the same "
                      "passes; includes the 9 B/block of widths[] / offsets[]")
    ranks = [{"rank": r, "first_block": block_range(CONFIG5_BLOCKS, world, r)[0], "blocks": int(v[0]),
              "kernel_ms_avg": round(v[1], 4), "GBps": round(v[2] / (v[1] / 1e3) / 1e9, 1),
              "frac": round(v[2] / (v[1] / 1e3) / 1e9 / HBM_PEAK_GBPS, 4)} for r, v in enumerate(per_rank)]
    for r, f in enumerate(flags):
        ranks[r]["correct"] = f
        ranks[r]["verified_blocks"] = verified[r]
    return {
        "metric": "billion integers/sec decoded (u32 mixed widths 1-32, 10 B-integer column sharded over the GPUs)",
        "workload": "unpack u32 width[b] = 1 + b mod 32, 9 765 625 blocks in total (BASELINE.json configs[4]); widths[] / "
                    "offsets[] device-resident, one launch per rank per step, contiguous block range per GPU, no collective",
        "value": round(CONFIG5_BLOCKS * 1024 * args.steps / elapsed / 1e9, 2),
        "unit": "Gint/s",
        "scaling": "strong",
        "n_gpus": world,
        "steps": args.steps,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "aggregate_GBps": round(sum(v[2] for v in per_rank) * args.steps / elapsed / 1e9, 1),
        "per_rank": ranks,
        "roofline_rank0": dict(roofline(w, kern_ms, traffic, source, bare), **({"placement_probe_GBps": probe} if probe else {})),
        "placement": placement_text(placed, probe, classes),
        "correctness": check_text(flags, n_checked, verified),
        "flags": flags,
    }


def main():
    args = parse()
    if args.pmc_child:
        return pmc_child(args)
    if args.nccl_probe_child:
        return nccl_probe_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; running {world} ranks", file=sys.stderr)
    if args.single_device:
        local_rank = 0
    if args.dry_run:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            sys.exit("bench.py needs a GPU (there is no CPU path); --dry-run only exercises the launcher")
        if local_rank >= torch.cuda.device_count():
            sys.exit(f"rank {rank}: cuda:{local_rank} does not exist ({torch.cuda.device_count()} devices visible)")
        dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(dev)
    ctl = Control(args, world, rank, dev)
    control = {"control_backend": ctl.backend}
    if ctl.fallback_reason:
        control["control_fallback_reason"] = ctl.fallback_reason

    ty, width, op, _ = WORKLOADS[args.workload]
    strong_main = op == "unpack_mixed" and args.blocks == 10_000_000
    n, first = args.blocks, rank * args.blocks
    if strong_main:   # BASELINE.json configs[4]: 10 B integers in total, sharded by block range
        from fastlanes_amd.sharding import block_range
        first, n = block_range(CONFIG5_BLOCKS, world, rank)

    if args.dry_run:
        # plumbing only: the same barrier / max-reduce / gather calls, no codec work, no number
        t0 = time.perf_counter()
        ctl.barrier()
        ctl.all_max(time.perf_counter() - t0)
        per_rank = ctl.gather([float(rank), float(n)])
        flags = [bool(v[0]) for v in ctl.gather([0.0 if rank == args.inject_mismatch else 1.0])]   # the verdict plumbing
        c5 = None if (args.no_config5 or strong_main) else config5_leg(args, world, rank, dev, ctl)
        if rank == 0:
            print(json.dumps(dict({"metric": "DRY RUN (no GPU work): launcher / rendezvous / sharding plumbing only",
                                   "value": None, "unit": "Gint/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                                   "dry_run": True, "ranks": [int(v[0]) for v in per_rank],
                                   "blocks_per_rank": [int(v[1]) for v in per_rank], "config5_strong": c5,
                                   "per_rank": [{"rank": r, "correct": f} for r, f in enumerate(flags)],
                                   "correctness": check_text(flags, 0)}, **control)), flush=True)
        ctl.close()
        if not all(flags):
            sys.exit(1)
        return

    import fastlanes_amd as fl
    fl.load()  # fails loudly if the HIP extension is missing

    w, probe = placed_workload(args.workload, n, first, rank, dev, args, args.single_device)
    elapsed, kern_ms = timed_region(w.step, args, ctl)
    avg_ms = sum(kern_ms) / len(kern_ms)
    per_rank = ctl.gather([float(n), avg_ms, float(w.bytes)])
    flags, n_checked, verified = run_check(w, args, ctl)       # every rank, its own slice, outside the timed region
    placed, classes = w.placement, getattr(w, "classes", "")
    bare = bare_stream(w) if rank == 0 else None     # after the checks (it overwrites the output), same buffers, same run
    tune = None
    if rank == 0 and world == 1 and not args.no_dispatch_check:
        try:
            tune = dispatch_check(w, gib=8.0)
        except Exception as e:                       # measurement tooling must never take the bench line down
            print(f"dispatch_check failed: {e!r}", file=sys.stderr)
    release(w)                                       # leg 1 is measured and checked: its column can go

    # ---- rank 0, N=1: the cpu_baseline leg (with the checks, the only place bench.py touches oracle/)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, ty, width, op)

    out = None
    if rank == 0:
        ints = CONFIG5_BLOCKS * 1024 if strong_main else n * 1024 * world
        value = ints * args.steps / elapsed / 1e9
        traffic = traffic_source = None
        if world == 1 and not args.no_pmc:
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
        if args.workload == "u32_w7_unpack":
            workload = f"{op} {ty} W={width}, {n} blocks x 1024 values per GPU (BASELINE.json configs[1])"
        elif op == "unpack_mixed":
            pattern = ("seeded-random in 1..32 (the variant SURVEY.md 8(d) asks for next to BASELINE.json configs[4])" if "random" in args.workload
                       else "1 + b mod 32 (BASELINE.json configs[4])")
            workload = (f"{op} {ty} width[b] = {pattern}, widths[]/offsets[] device-resident, "
                        f"{n} blocks on rank 0" + (f" of {CONFIG5_BLOCKS} in total" if strong_main else " per GPU"))
        else:
            workload = f"{op} {ty} W={width}, {n} blocks per GPU"
        ranks = [{"rank": r, "blocks": int(v[0]), "kernel_ms_avg": round(v[1], 4),
                  "GBps": round(v[2] / (v[1] / 1e3) / 1e9, 1)} for r, v in enumerate(per_rank)]
        for r, f in enumerate(flags):
            ranks[r]["correct"] = f
            ranks[r]["verified_blocks"] = verified[r]
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
            "scaling": "strong" if strong_main else "weak",
            "vs_baseline": None,
            "dtype": ty,
            "data": "synthetic (uniform random packed bits, generated on device; inputs resident in HBM)",
            "config": {"workload": workload, "blocks_per_gpu": n, "sharding": "contiguous block range per GPU, no collective"},
            "roofline": roofline(w, kern_ms, traffic, traffic_source, bare),
            "per_rank": ranks,
            "correctness": check_text(flags, n_checked, verified),
        }
        out["config"]["placement"] = placement_text(placed, probe, classes)
        if probe:
            out["roofline"]["placement_probe_GBps"] = probe
        out.update(control)
        if tune is not None:
            out["dispatch_check"] = tune
        if cpu is not None:
            out["cpu_baseline"] = cpu

    # ---- second leg: BASELINE.json configs[4] strong-scaled over the same ranks ------------------
    bad = not all(flags)
    if not args.no_config5 and not strong_main:
        c5 = config5_leg(args, world, rank, dev, ctl)
        bad = bad or not all(c5.pop("flags"))
        if rank == 0:
            out["config5_strong"] = c5

    if rank == 0:
        print(json.dumps(out), flush=True)
    ctl.close()
    rc = 1 if bad else 0            # every rank knows every rank's verdict: all of them fail together
    if ctl.inprocess_failed:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(rc)
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
