#!/usr/bin/env python3
"""Regenerates tests/golden/golden.json from the CPU oracle.

    python tests/golden/make_golden.py

The reference crate ships no golden byte vectors and cannot be executed here
(no Rust toolchain), so these vectors are ORACLE-derived: inputs come from the
counter-based generator in tests/datagen.py, expected outputs are SHA-256
digests of the oracle's little-endian output bytes.  The oracle itself is
pinned separately (tests/test_oracle_reference_tests.py).  The fixture lets
the GPU parity tests run against committed data, independent of oracle/ at
run time, and makes any future oracle change visible as a diff.  Committed
digests are immutable: the script refuses to overwrite one that changed.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np  # noqa: E402

from datagen import sha, values  # noqa: E402
from oracle_lib import lanes, load_oracle, packed_len, tbits  # noqa: E402

N_BLOCKS = 3


def case_inputs(ty, w):
    """The exact inputs every consumer of golden.json regenerates."""
    T = tbits(ty)
    seed = 1000 * T + w
    return {
        "values": values(ty, N_BLOCKS * 1024, seed),                        # full width (over-wide)
        "packed": values(ty, N_BLOCKS * packed_len(ty, w), seed + 100000),  # arbitrary packed bits
        "refs": values(ty, N_BLOCKS, seed + 200000),
        "bases": values(ty, N_BLOCKS * lanes(ty), seed + 300000),
    }


def main():
    o = load_oracle()
    out = {"n_blocks": N_BLOCKS, "generator": "tests/datagen.py splitmix64", "cases": {}}
    for ty in ("u8", "u16", "u32", "u64"):
        T = tbits(ty)
        for w in range(T + 1):
            i = case_inputs(ty, w)
            out["cases"][f"{ty}/{w}"] = {
                "pack": sha(o.batch("pack", ty, w, i["values"])),
                "unpack": sha(o.batch("unpack", ty, w, i["packed"], n_blocks=N_BLOCKS)),
                "for_pack": sha(o.batch("for_pack", ty, w, i["values"], aux=i["refs"])),
                "unfor_pack": sha(o.batch("unfor_pack", ty, w, i["packed"], aux=i["refs"], n_blocks=N_BLOCKS)),
                "undelta_pack": sha(o.batch("undelta_pack", ty, w, i["packed"], aux=i["bases"], n_blocks=N_BLOCKS)),
            }
        i = case_inputs(ty, T)
        out["cases"][f"{ty}/misc"] = {
            "delta": sha(o.batch("delta", ty, None, i["values"], aux=i["bases"])),
            "undelta": sha(o.batch("undelta", ty, None, i["values"], aux=i["bases"])),
            "transpose": sha(o.batch("transpose", ty, None, i["values"])),
            "untranspose": sha(o.batch("untranspose", ty, None, i["values"])),
        }
    # unpack_single (bitpacking.rs:132-179): the closed-form reader, EVERY index of two blocks per (T, W)
    # -- the matrix of the reference's try_round_trip (bitpacking.rs:273-315)
    out["unpack_single"] = {}
    for ty in ("u8", "u16", "u32", "u64"):
        T = tbits(ty)
        dt = values(ty, 1, 0).dtype
        for w in range(T + 1):
            pl = packed_len(ty, w)
            pk = values(ty, 2 * pl, 3300 + 64 * T + w)
            got = np.array([o.unpack_single(ty, w, pk[(i // 1024) * pl:(i // 1024 + 1) * pl], i % 1024)
                            for i in range(2048)], dtype=dt)
            out["unpack_single"][f"{ty}_w{w}"] = sha(got)
    # Known-answer vectors of SURVEY.md section 8(c) (independently model-derived there)
    out["survey_kats"] = {
        "KAT-2 u16 W=3 v[i]=i%8": "f949547d2b920f409dc21441e8ce7d412965a9ff3eac94d551362f689372db20",
        "KAT-3 u32 W=10 v[i]=i": "fded69a758643dbc59d8e5afc1cd28f96576f71c04aafbe0507dd7837e5a6d1c",
        "KAT-4 u32 W=7 v[i]=i&127": "16f02eec2ce2d18d6ac9cb51e5768981332865bf8cf2ba9fbf15712b48e59bb5",
        "KAT-5 u64 W=17 v[i]=(i*2654435761)&0x1FFFF": "6f2ff76d32f1ac12771043c4d884b8a7972fa1a6e5f15b18e9cd3ce3f8b510f5",
        "KAT-7 u16 W=9 delta bench": "7123aa8cd64fba3555abb4cf3180f8b273745ba6cf244314f7901bfcdf9db2a4",
    }
    # Committed vectors are IMMUTABLE: a regenerated digest that differs from the committed one means the
    # oracle changed behaviour -- that must fail loudly, not silently rewrite the fixture.  New keys may be
    # added; `--force` is for a deliberate, reviewed change only.
    path = os.path.join(HERE, "golden.json")
    if os.path.exists(path) and "--force" not in sys.argv:
        old = json.load(open(path))

        def changed(a, b, where):
            bad = []
            for k, v in a.items():
                if k not in b:
                    bad.append(f"{where}{k} (removed)")
                elif isinstance(v, dict):
                    bad += changed(v, b[k], f"{where}{k}/")
                elif v != b[k]:
                    bad.append(f"{where}{k}")
            return bad
        bad = changed(old, out, "")
        if bad:
            sys.exit(f"refusing to overwrite {len(bad)} committed golden vector(s): {bad[:8]} ... "
                     "(the oracle's output changed; rerun with --force only after reviewing why)")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote golden.json with", len(out["cases"]), "cases +", len(out["unpack_single"]), "unpack_single digests")


if __name__ == "__main__":
    main()
