mkdir -p gpurun_out
R=/root/repo/gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $R/pytest_gpu.log
timeout 600 python bench.py > $R/bench.json 2> $R/bench.err; echo "bench rc=$?"; cat $R/bench.json
rm -f $R/bench_other.jsonl
for wl in u64_w17_unpack u64_w17_pack u32_w12_undelta_pack u32_w7_pack u16_w3_unpack; do
  timeout 300 python bench.py --workload $wl --steps 10 --no-cpu-baseline >> $R/bench_other.jsonl 2>> $R/bench_other.err
done
python -c "
import sys, json
for l in open('$R/bench_other.jsonl'):
    d = json.loads(l); r = d['roofline']
    print(d['config']['workload'][:40], d['value'], 'Gint/s', r['achieved'], 'GB/s', r['frac'])
"
