# tools/gpu/final_check.sh <tag>: what the driver runs at round end, on one box: the GPU suite, smoke(), the default bench line
R=gpurun_out/r06c; mkdir -p $R; T=${1:-a}
( timeout 1500 python -m pytest tests -m gpu -q -x ) > $R/final_suite_$T.txt 2>&1; echo "suite rc=$?"; tail -n 2 $R/final_suite_$T.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
( time timeout 900 python bench.py > $R/final_bench_$T.json 2> $R/final_bench_$T.err ) 2>&1 | grep real
python - <<PY
import json,re
d=json.loads(open("$R/final_bench_$T.json").read().strip().splitlines()[-1])
m=re.search(r"input first: (\w+)", d["config"]["placement"])
r=d["roofline"]; c=d["config5_strong"]
print("headline", d["value"], r["frac"], r["frac_of_bare_stream"], r["placement_probe_GBps"], m.group(1) if m else "", "| config5", c["value"], c["roofline_rank0"]["frac"], "| cpu", d["cpu_baseline"]["value"])
PY
