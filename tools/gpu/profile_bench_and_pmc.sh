R=/root/repo/gpurun_out
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
rm -rf $R/prof_r01_trace $R/prof_r01_fetch $R/prof_r01_write
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_r01_trace -o bench -- python /root/repo/bench.py --steps 10 --no-cpu-baseline > $R/prof_bench.log 2>&1; echo "rocprof rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/prof_r01_fetch -o pmc -- python /root/repo/tools/pmc_probe.py > $R/prof_fetch.log 2>&1; echo "rocprof fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/prof_r01_write -o pmc -- python /root/repo/tools/pmc_probe.py > $R/prof_write.log 2>&1; echo "rocprof write rc=$?"
tail -2 $R/prof_bench.log
