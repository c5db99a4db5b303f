# round 4, seventh GPU call: full check alone (library loaded before torch: one HIP runtime), bench u64 workloads with probed placement, the whole GPU suite
R=gpurun_out/r04g
mkdir -p $R
( time timeout 1500 python -m pytest tests/test_gpu_full_check.py -m gpu -q -k under_load ) > $R/full_check_alone.txt 2>&1; echo "full check (under_load only) rc=$?"; tail -4 $R/full_check_alone.txt
rm -f $R/bench_other.jsonl
for wl in u64_w17_pack u64_w17_unpack u32_w12_undelta_pack; do
  timeout 600 python bench.py --workload $wl --steps 10 --no-cpu-baseline --no-config5 --no-pmc >> $R/bench_other.jsonl 2>> $R/bench_other.err; echo "$wl rc=$?"
done
python - <<'PY'
import json
for l in open("gpurun_out/r04g/bench_other.jsonl"):
    d = json.loads(l); print(d["config"]["workload"][:48], d["value"], d["roofline"]["frac"], d["roofline"].get("placement_probe_GBps"), d["per_rank"][0].get("verified_blocks"))
PY
tail -n 5 $R/bench_other.err
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $R/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 6 $R/pytest.txt
