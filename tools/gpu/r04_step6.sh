# round 4, sixth GPU call: full-output check (alone first: order independence), the known-bad build, then the bench with --placement auto / --verify full
R=gpurun_out/r04f
mkdir -p $R
( time timeout 1500 python -m pytest tests/test_gpu_full_check.py -m gpu -q -k under_load ) > $R/full_check_alone.txt 2>&1; echo "full check (under_load only) rc=$?"; tail -4 $R/full_check_alone.txt
( time timeout 1500 python -m pytest tests/test_gpu_full_check.py -m gpu -q ) > $R/full_check.txt 2>&1; echo "full check rc=$?"; tail -4 $R/full_check.txt
( time FL_LIB=$(pwd)/fastlanes_amd/libfastlanes_amd_badscan.so timeout 900 python -m pytest tests/test_gpu_full_check.py -m gpu -q -k "under_load" ) > $R/full_check_badscan.txt 2>&1; echo "badscan rc=$? (nonzero expected)"
grep -E "^(FAILED|ERROR)|passed|failed" $R/full_check_badscan.txt | head -20
timeout 900 python bench.py > $R/bench_u32w7.json 2> $R/bench.err; echo "bench rc=$?"
rm -f $R/bench_other.jsonl
for wl in u64_w17_pack u64_w17_unpack u32_w12_undelta_pack u32_w7_pack u16_w3_unpack; do
  timeout 400 python bench.py --workload $wl --steps 10 --no-cpu-baseline --no-config5 --no-pmc >> $R/bench_other.jsonl 2>> $R/bench_other.err; echo "$wl rc=$?"
done
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04f/bench_u32w7.json"))
print("HEADLINE", d["value"], d["roofline"]["frac"], d["roofline"].get("placement_probe_GBps"), d["correctness"][:150])
c = d["config5_strong"]; print("CONFIG5", c["value"], c["roofline_rank0"]["frac"], c["roofline_rank0"].get("placement_probe_GBps"), c["per_rank"][0].get("verified_blocks"))
for l in open("gpurun_out/r04f/bench_other.jsonl"):
    d = json.loads(l); print(d["config"]["workload"][:48], d["value"], d["roofline"]["frac"], d["roofline"].get("placement_probe_GBps"), d["per_rank"][0].get("verified_blocks"))
PY
tail -5 $R/bench.err $R/bench_other.err
