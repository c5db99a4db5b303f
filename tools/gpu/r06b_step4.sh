# after regenerating fl_dispatch_table.inc from constructed sweeps: suite, the BASELINE workloads, every (T, W) in constructed pairs
R=gpurun_out/r06c; mkdir -p $R
bash tools/gpu/stages.sh r06c suite
timeout 600 python bench.py > $R/bench_u32w7.json 2> $R/bench.err; echo "bench rc=$?"
rm -f $R/bench_other.jsonl
for wl in u32_mixed_unpack u64_w17_unpack u64_w17_pack u32_w12_undelta_pack u32_w7_pack u16_w3_unpack; do
  timeout 500 python bench.py --workload $wl --steps 10 --cpu-seconds 3 --no-config5 >> $R/bench_other.jsonl 2>> $R/bench_other.err; echo "$wl rc=$?"
done
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06c/bench_u32w7.json").read().strip().splitlines()[-1])
print("headline", d["roofline"]["frac"], "config5", d["config5_strong"]["roofline_rank0"]["frac"], d.get("dispatch_check",{}).get("table_behind_pct"))
for l in open("gpurun_out/r06c/bench_other.jsonl"):
    d=json.loads(l); print(d["config"]["workload"][:40], d["roofline"]["frac"], d.get("dispatch_check",{}).get("table_behind_pct"))
PY
bash tools/gpu/allwidths_constructed.sh $R/sweep_allwidths_constructed.txt | grep -v "eight"
