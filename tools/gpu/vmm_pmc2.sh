#!/bin/bash
# tools/gpu/vmm_pmc2.sh <outdir>: UTCL1 translation counters of the unpack kernel on a "bad" and a "good" address range of the same process
# (tools/exp_vmm layouts mode: layouts at the first address ranges, then the same chunks at fresh ranges)
R=$PWD; O=$R/${1:-gpurun_out/r06}/vmm_pmc2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export EXP_KEEP_VA=1 EXP_SEP_LATE=1 EXP_NO_STREAM=1 EXP_FRESH_VA_AFTER=1
for i in 1 2; do
  timeout 400 rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum --output-format csv -d $O/p$i -o p$i -- $R/tools/exp_vmm 100 1024 unpack32w7 10000000 1 > $O/p$i.log 2>&1
  echo "pass $i rc=$?"
done
