#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03; mkdir -p $O
for c in "compare u32 20" "compare u32 7" "undelta_pack u32 12" "undelta_pack u16 9"; do timeout 200 python tools/exp_thin_stream.py $c; done > $O/exp_thin_stream.txt 2>&1
cat $O/exp_thin_stream.txt
