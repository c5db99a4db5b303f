set -x
mkdir -p gpurun_out
R=/root/repo/gpurun_out
for wl in u64_w17_unpack u64_w17_pack u32_w12_undelta_pack u32_w7_pack u16_w3_unpack; do
  timeout 300 python bench.py --workload $wl --steps 10 --no-cpu-baseline >> $R/bench_other.jsonl 2>> $R/bench_other.err
done
cat $R/bench_other.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print(d['config']['workload'][:40], d['value'], 'Gint/s', r['achieved'], 'GB/s', r['frac'])
"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_r01_trace -o bench -- python /root/repo/bench.py --steps 10 --no-cpu-baseline > $R/prof_bench.log 2>&1; echo "rocprof rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/prof_r01_fetch -o pmc -- python /root/repo/tools/pmc_probe.py > $R/prof_fetch.log 2>&1; echo "rocprof fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/prof_r01_write -o pmc -- python /root/repo/tools/pmc_probe.py > $R/prof_write.log 2>&1; echo "rocprof write rc=$?"
ls -laR $R/prof_r01_trace $R/prof_r01_fetch $R/prof_r01_write | head -40
