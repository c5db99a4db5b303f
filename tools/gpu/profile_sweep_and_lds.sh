R=/root/repo/gpurun_out
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -iE "SQ_LDS|SQ_WAVES|SQ_INSTS_VALU |SQ_BUSY_CYCLES|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_VALU|SQ_INSTS_LDS" | head -20 > $R/pmc_names.txt
rm -rf $R/prof_r01_sweep $R/prof_r01_lds
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_r01_sweep -o sweep -- python /root/repo/tools/sweep.py --cases quick --gb 8 --reps 3 > $R/prof_sweep.log 2>&1; echo "rc=$?"
timeout 900 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d $R/prof_r01_lds -o lds -- python /root/repo/tools/sweep.py --cases orig --gb 4 --reps 2 > $R/prof_lds.log 2>&1; echo "rc=$?"
ls $R/prof_r01_sweep $R/prof_r01_lds; tail -3 $R/prof_lds.log
