# SQ / LDS counters of the headline kernel (and the other bench kernels): where do wave cycles go?
R=/root/repo/gpurun_out
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
rm -rf $R/prof_r01_sq $R/prof_r01_sq2
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $R/prof_r01_sq -o sq -- python /root/repo/tools/pmc_probe.py > $R/prof_sq.log 2>&1; echo "rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d $R/prof_r01_sq2 -o sq2 -- python /root/repo/tools/pmc_probe.py > $R/prof_sq2.log 2>&1; echo "rc=$?"
tail -2 $R/prof_sq2.log
