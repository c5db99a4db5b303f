#!/usr/bin/env python3
"""HBM bytes actually FETCHED per lookup of the batched unpack_single (a7: bitpacking.rs:132-200), by index pattern -- what bounds
random access.  Re-runs itself under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (counters in their own pass, no other trace domain;
KiB units; gfx950 reports about half of wide coalesced reads, so the factor is taken from a copy of known size in the same pass, as
MI355X_MICROARCH.md prescribes and bench.py does) and prints bytes per lookup next to the lookup's own need (1-2 words of T) and the
32-byte sector.        python tools/pmc_single.py  >  profiles/<round>_pmc_unpack_single.txt"""
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CAL_BYTES = 2 << 30
N_BLOCKS, K = 1_000_000, 64_000_000
CASES = (("u32", 7), ("u64", 17), ("u16", 3), ("u8", 3))
PATTERNS = ("random", "sorted", "strided (one per block)", "dense (whole blocks in order)")


def child():
    import torch
    import fastlanes_amd as fl
    from bench import rand_u8
    dev = torch.device("cuda:0")
    lib = fl.load()
    a = rand_u8(CAL_BYTES, 5, dev)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    del a, b
    TDT = {"u8": torch.uint8, "u16": torch.uint16, "u32": torch.uint32, "u64": torch.uint64}
    for ty, w in CASES:
        pk = rand_u8(N_BLOCKS * 128 * w, 1, dev).view(TDT[ty])
        g = torch.Generator(device=dev)
        g.manual_seed(5)
        idx = torch.randint(0, N_BLOCKS * 1024, (K,), dtype=torch.int64, device=dev, generator=g)
        ar = torch.arange(K, dtype=torch.int64, device=dev)
        pats = (idx, torch.sort(idx).values, (ar % N_BLOCKS) * 1024 + (ar * 7) % 1024, ar)
        out = torch.empty(K, dtype=TDT[ty], device=dev)
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        for ii in pats:
            for _ in range(2):
                assert getattr(lib, f"fl_{ty}_unpack_single")(w, pk.data_ptr(), N_BLOCKS, ii.data_ptr(), K, out.data_ptr(), err.data_ptr(), None) == 0
            torch.cuda.synchronize()
        del pk, idx, ar, pats, out


def main():
    if "--child" in sys.argv:
        return child()
    if shutil.which("rocprofv3") is None:
        sys.exit("rocprofv3 not found")
    d = tempfile.mkdtemp(prefix="fl_pmc_single_", dir="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", "FETCH_SIZE", "--output-format", "csv", "-d", d, "-o", "pmc", "--",
           sys.executable, os.path.abspath(__file__), "--child"]
    subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    rows = list(csv.DictReader(open(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0])))
    rows = [r for r in rows if r["Counter_Name"] == "FETCH_SIZE"]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    cal = [float(r["Counter_Value"]) * 1024 for r in rows
           if float(r["Counter_Value"]) * 1024 > 0.25 * CAL_BYTES and ("copy" in r["Kernel_Name"].lower() or "elementwise" in r["Kernel_Name"].lower())
           and "distribution" not in r["Kernel_Name"]]
    factor = CAL_BYTES / (sum(cal) / len(cal))
    singles = [float(r["Counter_Value"]) * 1024 for r in rows if "k_unpack_single" in r["Kernel_Name"]]
    assert len(singles) == 2 * len(PATTERNS) * len(CASES), len(singles)
    print(f"# HBM bytes fetched per lookup of fl_<ty>_unpack_single ({K} lookups into a {N_BLOCKS}-block column; rocprofv3 --pmc FETCH_SIZE, KiB units, "
          f"x{factor:.3f} from a {CAL_BYTES >> 30}-GiB copy in the same pass).  Every lookup also reads its 8-byte index (sequentially) and needs 1-2 words "
          "of T from the column (bitpacking.rs:164-178).  READING: the factor is what WIDE COALESCED reads need on gfx950 (the copy, the dense pattern); a scattered "
          "lookup is one 64-byte request per word touched, and there the RAW counter is the plausible one: 8 B of index + 64 B x (1 + the share of fields that "
          "straddle two words).")
    k = 0
    for ty, w in CASES:
        esz = {"u8": 1, "u16": 2, "u32": 4, "u64": 8}[ty]
        for p in PATTERNS:
            raw = (singles[k] + singles[k + 1]) / 2 / K
            k += 2
            print(f"{ty:4s} W={w:<2d} {p:32s} raw counter {raw:6.1f} B per lookup, x{factor:.1f} = {raw * factor:6.1f} B  (the lookup needs its 8-byte index + {esz}-{2 * esz} B; "
                  f"column = {N_BLOCKS * 128 * w / 1e6:.0f} MB)")
    shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
