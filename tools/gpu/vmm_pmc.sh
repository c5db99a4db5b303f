#!/bin/bash
# tools/gpu/vmm_pmc.sh <outdir>: tools/exp_vmm's layouts mode under rocprofv3 --pmc (own passes): translation and DRAM-credit counters of the
# unpack kernel on the VMM-mapped layouts (slow on a pristine device) against the two hipMallocs at the end (fast)
R=$PWD; O=$R/${1:-gpurun_out/r06}/vmm_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export EXP_KEEP_VA=1 EXP_SEP_LATE=1 EXP_NO_STREAM=1
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
           "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum" \
           "GRBM_UTCL2_BUSY GRBM_EA_BUSY GRBM_GUI_ACTIVE TCC_TAG_STALL_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p$i -- $R/tools/exp_vmm 180 1024 unpack32w7 10000000 1 > $O/p$i.log 2>&1
  echo "pass $i rc=$?"
done
