#!/usr/bin/env python3
"""Static instruction mix of the kernels in a device assembly file (hipcc --cuda-device-only -S): VALU / LDS / VMEM / SALU / waits.
    python tools/kernel_instructions.py build/uint8_t_6.s [substring of the demangled name]
Static counts (loops and branches not weighed): a first look before SQ counters on the GPU."""
import collections
import re
import subprocess
import sys

lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
for i, name in starts:
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if want not in d:
        continue
    c = collections.Counter()
    for l in lines[i + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        t = l.strip()
        if not l.startswith("\t") or not t or t[0] in ".;":
            continue
        op = t.split()[0]
        k = ("valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "wait" if op.startswith("s_waitcnt") else
             "branch" if op.startswith(("s_cbranch", "s_branch")) else "salu" if op.startswith("s_") else
             "vmem" if op.startswith(("buffer_", "global_", "flat_")) else "other")
        c[k] += 1
        if k == "lds":
            c[op] += 1
    print(d[:100], dict(c))
