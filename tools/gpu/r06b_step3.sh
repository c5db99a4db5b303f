R=gpurun_out/r06b; mkdir -p $R
bash tools/gpu/stages.sh r06b suite
timeout 600 python bench.py > $R/bench_u32w7_nt.json 2> $R/bench_u32w7_nt.err; echo "bench rc=$?"
for wl in u64_w17_unpack u32_w12_undelta_pack u16_w3_unpack u32_w7_pack; do timeout 600 python bench.py --workload $wl > $R/bench_${wl}_nt.json 2> $R/bench_${wl}_nt.err; echo "$wl rc=$?"; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06b/bench_*_nt.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    r=d["roofline"]; print(d["config"]["workload"][:44], r["frac"], r.get("frac_of_bare_stream"), r.get("placement_probe_GBps"), d.get("dispatch_check",{}).get("table_behind_pct"), d.get("dispatch_check",{}).get("best_alternative"))
    if "config5_strong" in d: print("   config5", d["config5_strong"]["roofline_rank0"]["frac"])
PY
