R=gpurun_out/r06b; mkdir -p $R
bash tools/gpu/stages.sh r06b suite
for i in 1 2 3; do timeout 600 python bench.py --workload u64_w17_pack --no-cpu-baseline --no-pmc --no-config5 --verify sample > $R/bench_p64_$i.json 2> $R/bench_p64_$i.err; python - <<PY
import json,re
d=json.loads(open("$R/bench_p64_$i.json").read().strip().splitlines()[-1])
m=re.search(r"input first: (\w+)", d["config"]["placement"])
r=d["roofline"]; print("pack u64 W=17 run $i", r["frac"], r.get("frac_of_bare_stream"), r.get("placement_probe_GBps"), (m.group(1)[-19:] if m else ""))
PY
done
