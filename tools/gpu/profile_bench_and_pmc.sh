# Round-evidence run: bench (all workloads), rocprofv3 kernel trace of the bench, PMC passes.
set -x
R=/root/repo/gpurun_out
mkdir -p $R
cd /root/repo
timeout 600 python bench.py > $R/bench.json 2> $R/bench.err; echo "bench rc=$?"
rm -f $R/bench_other.jsonl
for wl in u64_w17_unpack u64_w17_pack u32_w12_undelta_pack u32_w7_pack u16_w3_unpack u32_mixed_unpack; do
  timeout 300 python bench.py --workload $wl --steps 10 --no-cpu-baseline >> $R/bench_other.jsonl 2>> $R/bench_other.err
done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/prof_r01_trace $R/prof_r01_fetch $R/prof_r01_write
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_r01_trace -o bench -- python /root/repo/bench.py --steps 10 --no-cpu-baseline --no-pmc > $R/prof_bench.log 2>&1; echo "rocprof rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/prof_r01_fetch -o pmc -- python /root/repo/tools/pmc_probe.py > $R/prof_fetch.log 2>&1; echo "rocprof fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/prof_r01_write -o pmc -- python /root/repo/tools/pmc_probe.py > $R/prof_write.log 2>&1; echo "rocprof write rc=$?"
