#!/usr/bin/env python3
"""VGPRs / spills / scratch / LDS of the kernels in a device assembly file (hipcc --cuda-device-only -S), demangled with c++filt.
    python tools/kernel_resources.py build/uint8_t_6.s [substring]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
for b in s.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if want not in d:
        continue
    f = lambda k: re.search(r"\.%s:\s+(\d+)" % k, b).group(1)
    print(f"{d[:140]:140s} vgpr {f('vgpr_count'):>3s} sgpr {f('sgpr_count'):>3s} spill {f('vgpr_spill_count')} scratch {f('private_segment_fixed_size')}")
